// In-wave and cross-wave overlap of VALU with fp32 MFMA on one SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: in-wave, dependent chain: [1 MFMA, NV v_fma] x
// MODE 1: in-wave, MFMA only (dependent chain)
// MODE 2: in-wave, VALU only (NV per iteration)
// MODE 3: cross-wave: waves 0-3 dependent MFMA chain, waves 4-7 VALU
// MODE 4: as 3 with s_setprio 3 on the VALU waves
// MODE 5: as 3 with s_setprio 3 on the MFMA waves
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters, int nthr) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  float a = threadIdx.x * 0.5f, b = 1.0001f;
  if ((int)threadIdx.x >= nthr) return;
  const bool mf = MODE == 0 || MODE == 1 || (MODE >= 3 && wave < 4);
  const bool va = MODE == 0 || MODE == 2 || (MODE >= 3 && wave >= 4);
  if (MODE == 4 && wave >= 4) __builtin_amdgcn_s_setprio(3);
  if (MODE == 5 && wave < 4) __builtin_amdgcn_s_setprio(3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (mf) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      if (va) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], b, a);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 1.2345e-30f) out[threadIdx.x] = s;
}

template <int MODE, int NV>
float run(float* d, int iters, int nthr) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(512), 0, 0, d, iters, nthr);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(512), 0, 0, d, iters, nthr);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}

int main() {
  float* d; (void)hipMalloc(&d, 4096);
  const int it = 4000;  // x 8 MFMAs (64 cycles each) = 2.05 M cycles = ~0.9 ms
  printf("one wave per SIMD (256 threads):\n");
  printf("  MFMA chain only                         : %8.1f us\n", run<1, 0>(d, it, 256));
  printf("  VALU only, 8 / 16 / 24 / 32 per slot    : %8.1f %8.1f %8.1f %8.1f us\n", run<2, 8>(d, it, 256), run<2, 16>(d, it, 256), run<2, 24>(d, it, 256), run<2, 32>(d, it, 256));
  printf("  [MFMA + n VALU] interleaved in ONE wave : %8.1f %8.1f %8.1f %8.1f us\n", run<0, 8>(d, it, 256), run<0, 16>(d, it, 256), run<0, 24>(d, it, 256), run<0, 32>(d, it, 256));
  printf("two waves per SIMD (512 threads): waves 0-3 MFMA chain, waves 4-7 VALU (n per MFMA slot)\n");
  printf("  n = 8 / 16 / 24 / 32                    : %8.1f %8.1f %8.1f %8.1f us\n", run<3, 8>(d, it, 512), run<3, 16>(d, it, 512), run<3, 24>(d, it, 512), run<3, 32>(d, it, 512));
  printf("  same, VALU waves at s_setprio 3         : %8.1f %8.1f %8.1f %8.1f us\n", run<4, 8>(d, it, 512), run<4, 16>(d, it, 512), run<4, 24>(d, it, 512), run<4, 32>(d, it, 512));
  printf("  same, MFMA waves at s_setprio 3         : %8.1f %8.1f %8.1f %8.1f us\n", run<5, 8>(d, it, 512), run<5, 16>(d, it, 512), run<5, 24>(d, it, 512), run<5, 32>(d, it, 512));
  return 0;
}
