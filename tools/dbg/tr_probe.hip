// Semantics probe of ds_read_b64_tr_b16 (gfx950): LDS holds element index e at element e; lane l supplies byte address 8 l.
// Prints, per lane, the four element indices it receives.   hipcc --offload-arch=gfx950 tools/dbg/tr_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(sm + 4 * l));
  for (int j = 0; j < 4; ++j) out[4 * l + j] = v[j];
}
int main() {
  short* d;
  short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], l % 4 == 3 ? "\n" : "   ");
  return 0;
}
