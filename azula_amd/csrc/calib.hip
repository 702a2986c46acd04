// Known-traffic micro-kernels for calibrating the rocprofv3 HBM counters (FETCH_SIZE / WRITE_SIZE) on gfx950.
// MI355X_MICROARCH.md (section HBM): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read and "other
// access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  These
// kernels touch every byte of a buffer much larger than the 256 MiB Infinity Cache exactly ONCE, in the access shapes of
// the product kernels (16 B/lane streams; 8 B/lane patch gathers that hit a cache line as four 32-byte sectors from
// four consecutive instructions; 16 B/lane row stores), so bytes / counter is the correction factor for that shape.
// Measurement support only: nothing on the sampling path calls them (tools/pmc_traffic.py, bench.py's roofline leg).
#include "common.h"

namespace {

// Rows of `row_bytes`; a "sector" = `group_bytes` contiguous bytes read by group_bytes / W adjacent lanes; a block of
// 256 threads covers 256 / (group_bytes / W) rows per pass and walks the sectors of its rows one after the other
// (so the sibling sectors of a 128-byte line are requested by consecutive instructions of the same wave).
template <int W>
__global__ __launch_bounds__(256) void calib_read_kernel(const char* __restrict__ src, float* __restrict__ sink,
                                                         int64_t rows, int64_t row_bytes, int group_bytes) {
  const int lanes_per_group = group_bytes / W;
  const int rows_per_block = 256 / lanes_per_group;
  const int g = threadIdx.x % lanes_per_group;
  const int r_in = threadIdx.x / lanes_per_group;
  const int sectors = (int)(row_bytes / group_bytes);
  float acc = 0.f;
  for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block; r0 < rows; r0 += (int64_t)gridDim.x * rows_per_block) {
    const int64_t r = r0 + r_in;
    if (r >= rows) continue;
    const char* p = src + r * row_bytes + (int64_t)g * W;
    for (int s0 = 0; s0 < sectors; s0 += 4) {
      if (W == 8) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = s0 + u < sectors ? *reinterpret_cast<const float2*>(p + (int64_t)(s0 + u) * group_bytes) : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
      } else {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = s0 + u < sectors ? *reinterpret_cast<const float4*>(p + (int64_t)(s0 + u) * group_bytes)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      }
    }
  }
  if (acc == 1.2345e37f) sink[0] = acc;  // never true for the calibration data; keeps the loads alive
}

// 16 B/lane stores of rows of `row_bytes` (the conv epilogues store 256- or 512-byte pixel rows with 16 / 32 lanes).
__global__ __launch_bounds__(256) void calib_write_kernel(float* __restrict__ dst, int64_t n16, float value) {
  const float4 v = make_float4(value, value, value, value);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(dst)[i] = v;
}

// Matrix pipe only: every wave issues `iters` x 8 independent v_mfma_f32_32x32x2_f32 on registers -- no memory, no LDS, no
// vector work.  What this sustains is the chip's fp32 MFMA rate under its power limit (the guide's 157.3 TF/s is 256 CUs x
// 256 FLOP per cycle at 2.4 GHz; under a dense MFMA load the clock settles near 2.0 GHz).
typedef float f32x16c __attribute__((ext_vector_type(16)));
// RANDOM = false: every MFMA of a lane multiplies the same two values (a0 scaled by the lane, b0) -- the multiplier inputs and
// operand buses barely toggle, and the chip draws ~690 W.  RANDOM = true: 8 + 8 per-lane pseudo-random operands in [-1, 1),
// a different pair for each of the 8 accumulator chains, so consecutive MFMAs (and the 16 passes inside one) see unrelated
// bit patterns as a real GEMM does: the power the matrix pipe draws on real data is what caps the clock (tools/power_probe.py).
template <bool RANDOM>
__global__ __launch_bounds__(256) void calib_mfma_kernel(float* sink, int iters, float a0, float b0) {
  f32x16c acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  float a[8], b[8];
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    if (RANDOM) {
      h = h * 1664525u + 1013904223u;
      a[f] = a0 * ((float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f);
      h = h * 1664525u + 1013904223u;
      b[f] = b0 * ((float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f);
    } else {
      a[f] = a0 * (1.0f + (float)(threadIdx.x & 7) * 0.125f);
      b[f] = b0;
    }
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f], b[f], acc[f], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int f = 0; f < 8; ++f) t += acc[f][0] + acc[f][15];
  if (t == -1.2345e37f) sink[0] = t;  // (never true: keeps the accumulators alive)
}

// The same on the bf16 pipe: `iters` x 8 independent v_mfma_f32_32x32x16_bf16 (the instruction of the bf16x3 kernels) on
// per-lane pseudo-random bf16 operands (8 + 8 values per lane and chain).  2516.8 TF/s nominal; what this sustains is the rate
// the 1400 W cap leaves the bf16 pipe on real data.
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void calib_mfma_bf16_kernel(float* sink, int iters, float a0, float b0) {
  f32x16c acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  union Frag {
    bf16x8c v;
    unsigned u[4];
  } a[8], b[8];
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  auto pair = [&](float s) {  // two bf16 values in [-s, s): the high halves of two random floats
    h = h * 1664525u + 1013904223u;
    const float x = s * ((float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f);
    h = h * 1664525u + 1013904223u;
    const float y = s * ((float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f);
    return (__builtin_bit_cast(unsigned, y) & 0xFFFF0000u) | (__builtin_bit_cast(unsigned, x) >> 16);
  };
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[f].u[j] = pair(a0);
      b[f].u[j] = pair(b0);
    }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f].v, b[f].v, acc[f], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int f = 0; f < 8; ++f) t += acc[f][0] + acc[f][15];
  if (t == -1.2345e37f) sink[0] = t;  // (never true: keeps the accumulators alive)
}

}  // namespace

extern "C" {

int az_calib_mfma_random_bf16(float* sink, int32_t workgroups, int32_t iters, float a, float b, az_stream_t stream) {
  AZ_REQUIRE(sink, AZ_E_NULL);
  AZ_REQUIRE(workgroups > 0 && iters > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(calib_mfma_bf16_kernel, dim3((unsigned)workgroups), dim3(256), 0, az_s(stream), sink, iters, a, b);
  return az_launch_status();
}

int az_calib_mfma_f32(float* sink, int32_t workgroups, int32_t iters, float a, float b, az_stream_t stream) {
  AZ_REQUIRE(sink, AZ_E_NULL);
  AZ_REQUIRE(workgroups > 0 && iters > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(calib_mfma_kernel<false>, dim3((unsigned)workgroups), dim3(256), 0, az_s(stream), sink, iters, a, b);
  return az_launch_status();
}

int az_calib_mfma_random_f32(float* sink, int32_t workgroups, int32_t iters, float a, float b, az_stream_t stream) {
  AZ_REQUIRE(sink, AZ_E_NULL);
  AZ_REQUIRE(workgroups > 0 && iters > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(calib_mfma_kernel<true>, dim3((unsigned)workgroups), dim3(256), 0, az_s(stream), sink, iters, a, b);
  return az_launch_status();
}

int az_calib_read_f32(const float* src, float* sink, int64_t nbytes, int32_t width, int32_t group_bytes,
                      int64_t row_bytes, az_stream_t stream) {
  AZ_REQUIRE(src && sink, AZ_E_NULL);
  AZ_REQUIRE((width == 8 || width == 16) && group_bytes >= width && group_bytes % width == 0, AZ_E_SHAPE);
  AZ_REQUIRE(256 % (group_bytes / width) == 0 && row_bytes >= group_bytes && row_bytes % group_bytes == 0, AZ_E_SHAPE);
  AZ_REQUIRE(nbytes > 0 && nbytes % row_bytes == 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(src), AZ_E_ALIGN);
  const int64_t rows = nbytes / row_bytes;
  const int rows_per_block = 256 / (group_bytes / width);
  int64_t grid = (rows + rows_per_block - 1) / rows_per_block;
  if (grid > 16384) grid = 16384;
  if (width == 8)
    hipLaunchKernelGGL(calib_read_kernel<8>, dim3((unsigned)grid), dim3(256), 0, az_s(stream), (const char*)src, sink,
                       rows, row_bytes, group_bytes);
  else
    hipLaunchKernelGGL(calib_read_kernel<16>, dim3((unsigned)grid), dim3(256), 0, az_s(stream), (const char*)src, sink,
                       rows, row_bytes, group_bytes);
  return az_launch_status();
}

int az_calib_write_f32(float* dst, int64_t nbytes, float value, az_stream_t stream) {
  AZ_REQUIRE(dst, AZ_E_NULL);
  AZ_REQUIRE(nbytes > 0 && nbytes % 16 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(dst), AZ_E_ALIGN);
  hipLaunchKernelGGL(calib_write_kernel, dim3(8192), dim3(256), 0, az_s(stream), dst, nbytes / 16, value);
  return az_launch_status();
}

}  // extern "C"
