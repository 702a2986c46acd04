// Does the gfx950 matrix pipe honour SUBNORMAL fp16 / bf16 operands?  (decides the domain of an fp16-piece split of fp32 operands)
//   hipcc --offload-arch=gfx950 tools/mfma_denorm_probe.hip -o tools/_build/mfma_denorm_probe && tools/_build/mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(float* out, unsigned short abits, unsigned short bbits, int kind) {
  u16x8 ua, ub;
  for (int i = 0; i < 8; ++i) { ua[i] = abits; ub[i] = bbits; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  if (kind == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ub), c, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
// conversions as the split would write them: which instructions does the compiler pick, and how do they round subnormal results?
__global__ void cvt_probe(float* out, float x0, float x1) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {x0, x1};
  const h2 h = __builtin_convertvector(v, h2);
  const f2 b = __builtin_convertvector(h, f2);
  out[0] = b.x; out[1] = b.y;
  out[2] = x0 - b.x; out[3] = x1 - b.y;
}
static float run(unsigned short a, unsigned short b, int kind) {
  float* d; hipMalloc(&d, 16);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, a, b, kind);
  float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); hipFree(d);
  return h;
}
int main() {
  // fp16: 0x0010 = 2^-20 (subnormal), 0x0001 = 2^-24 (smallest subnormal), 0x0400 = 2^-14 (smallest normal), 0x3C00 = 1.0, 0x7800 = 32768
  printf("f16 normal   2^-14 x 1.0   x16: %.9g (expect %.9g)\n", run(0x0400, 0x3C00, 0), 16 * 6.103515625e-05);
  printf("f16 subnorm  2^-20 x 1.0   x16: %.9g (expect %.9g if honoured)\n", run(0x0010, 0x3C00, 0), 16 * 9.5367431640625e-07);
  printf("f16 subnorm  2^-24 x 1.0   x16: %.9g (expect %.9g if honoured)\n", run(0x0001, 0x3C00, 0), 16 * 5.9604644775390625e-08);
  printf("f16 subnorm  2^-24 x 32768 x16: %.9g (expect %.9g if honoured)\n", run(0x0001, 0x7800, 0), 16 * 5.9604644775390625e-08 * 32768);
  printf("f16 subnorm  0x03FF x 1.0  x16: %.9g (expect %.9g if honoured)\n", run(0x03FF, 0x3C00, 0), 16 * 1023 * 5.9604644775390625e-08);
  printf("f16 sub x sub 2^-20 x 2^-20 x16: %.9g (expect %.9g if honoured)\n", run(0x0010, 0x0010, 0), 16 * 9.5367431640625e-07 * 9.5367431640625e-07);
  // bf16: 0x0040 = 2^-127 (subnormal), 0x7000 = 2^97
  printf("bf16 subnorm 2^-127 x 2^97 x16: %.9g (expect %.9g if honoured)\n", run(0x0040, 0x7000, 1), 16 * 9.313225746154785e-10);
  float* d; hipMalloc(&d, 16);
  const float xs[4][2] = {{1e-5f, 3e-6f}, {6.2e-5f, 5.9e-5f}, {1.0009765f, 70000.f}, {1e-8f, 2.9e-8f}};
  for (auto& x : xs) {
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(1), 0, 0, d, x[0], x[1]);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("cvt f16(%.9g, %.9g) = (%.9g, %.9g) residuals (%.9g, %.9g)\n", x[0], x[1], h[0], h[1], h[2], h[3]);
  }
  return 0;
}
