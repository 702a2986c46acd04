// Small-M linear layers (modulation / time-embedding MLPs: M = 1 or batch).  These are GEMVs:
// HBM-bound on the weight matrix, so each wave streams one weight row with 16-byte loads and
// reduces with wave64 shuffles; the activation row(s) stay in L1/L2.  Token-tensor linears
// (large M) go through the MFMA implicit-GEMM kernel in conv.hip instead.
#include "common.h"

namespace {

constexpr int ROWS = 4;  // activation rows handled per pass over a weight row

template <bool VEC>
__global__ __launch_bounds__(256) void linear_small_kernel(float* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias, int64_t M, int64_t N,
                                                           int64_t K, int in_act, int out_act) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= N) return;
  const float* w = W + n * K;
  for (int64_t m0 = (int64_t)blockIdx.y * ROWS; m0 < M; m0 += (int64_t)gridDim.y * ROWS) {
    float acc[ROWS] = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
      for (int64_t k = lane * 4; k < K; k += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          if (m0 + r < M) {
            float4 xv = *reinterpret_cast<const float4*>(x + (m0 + r) * ldx + k);
            if (in_act == 1) {
              xv.x = az_silu(xv.x);
              xv.y = az_silu(xv.y);
              xv.z = az_silu(xv.z);
              xv.w = az_silu(xv.w);
            }
            acc[r] = fmaf(xv.x, wv.x, acc[r]);
            acc[r] = fmaf(xv.y, wv.y, acc[r]);
            acc[r] = fmaf(xv.z, wv.z, acc[r]);
            acc[r] = fmaf(xv.w, wv.w, acc[r]);
          }
        }
      }
    } else {
      for (int64_t k = lane; k < K; k += 64) {
        const float wv = w[k];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          if (m0 + r < M) {
            float xv = x[(m0 + r) * ldx + k];
            if (in_act == 1) xv = az_silu(xv);
            acc[r] = fmaf(xv, wv, acc[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float s = az_wave_sum(acc[r]);
      if (lane == 0 && m0 + r < M) {
        float v = s + (bias ? bias[n] : 0.f);
        if (out_act == 1) v = az_silu(v);
        y[(m0 + r) * ldy + n] = v;
      }
    }
  }
}

// Grouped variant: blockIdx.y selects a group descriptor (device memory); all groups share M and the
// activations.  One launch replaces the per-block modulation MLPs of a backbone (they depend only on the
// conditioning vector, so they are hoisted to the front of the forward).
__global__ __launch_bounds__(256) void linear_small_grouped_kernel(const AzLinearGroup* __restrict__ groups, int64_t M,
                                                                   int in_act, int out_act) {
  const AzLinearGroup g = groups[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= g.N) return;
  const float* w = g.W + n * g.K;
  for (int64_t m0 = 0; m0 < M; m0 += ROWS) {
    float acc[ROWS] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t k = lane * 4; k < g.K; k += 256) {
      const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        if (m0 + r < M) {
          float4 xv = *reinterpret_cast<const float4*>(g.x + (m0 + r) * g.ldx + k);
          if (in_act == 1) {
            xv.x = az_silu(xv.x);
            xv.y = az_silu(xv.y);
            xv.z = az_silu(xv.z);
            xv.w = az_silu(xv.w);
          }
          acc[r] = fmaf(xv.x, wv.x, acc[r]);
          acc[r] = fmaf(xv.y, wv.y, acc[r]);
          acc[r] = fmaf(xv.z, wv.z, acc[r]);
          acc[r] = fmaf(xv.w, wv.w, acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float s = az_wave_sum(acc[r]);
      if (lane == 0 && m0 + r < M) {
        float v = s + (g.bias ? g.bias[n] : 0.f);
        if (out_act == 1) v = az_silu(v);
        g.y[(m0 + r) * g.ldy + n] = v;
      }
    }
  }
}

}  // namespace

extern "C" int az_linear_small_grouped_f32(const AzLinearGroup* groups_dev, int32_t ngroups, int32_t max_n, int64_t M,
                                           int32_t in_act, int32_t out_act, az_stream_t stream) {
  AZ_REQUIRE(groups_dev, AZ_E_NULL);
  AZ_REQUIRE(ngroups > 0 && ngroups <= 65535 && max_n > 0 && M > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(linear_small_grouped_kernel, dim3((unsigned)((max_n + 3) / 4), (unsigned)ngroups), dim3(256), 0,
                     az_s(stream), groups_dev, M, in_act, out_act);
  return az_launch_status();
}

extern "C" int az_linear_small_f32(float* y, int64_t ldy, const float* x, int64_t ldx, const float* W,
                                   const float* bias, int64_t M, int64_t N, int64_t K, int32_t in_act,
                                   int32_t out_act, az_stream_t stream) {
  AZ_REQUIRE(y && x && W, AZ_E_NULL);
  AZ_REQUIRE(M > 0 && N > 0 && K > 0 && ldy >= N && ldx >= K, AZ_E_SHAPE);
  const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && AZ_ALIGNED16(x) && AZ_ALIGNED16(W);
  dim3 grid((unsigned)((N + 3) / 4), (unsigned)((M + ROWS - 1) / ROWS > 64 ? 64 : (M + ROWS - 1) / ROWS));
  if (vec)
    hipLaunchKernelGGL(linear_small_kernel<true>, grid, dim3(256), 0, az_s(stream), y, ldy, x, ldx, W, bias, M, N, K,
                       in_act, out_act);
  else
    hipLaunchKernelGGL(linear_small_kernel<false>, grid, dim3(256), 0, az_s(stream), y, ldy, x, ldx, W, bias, M, N,
                       K, in_act, out_act);
  return az_launch_status();
}
