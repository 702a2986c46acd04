// A SLICE of a standard-normal draw, bit-identical to the same elements of the full tensor drawn by torch.randn / Tensor.normal_
// on this device (multi-GPU sampling, SURVEY 8e: every rank keeps the single-device random stream without drawing the whole
// batch).  ATen's kernel (ATen/native/cuda/DistributionTemplates.h: distribution_elementwise_grid_stride_kernel, unroll 4) gives
// thread idx of a launch of T = 256 * grid threads the Philox subsequence idx, and its j-th curand_normal4 call produces the
// elements idx + T (4 j + i), i = 0 .. 3 -- so element e comes from subsequence e mod T, call (e / T) / 4, component (e / T) mod 4,
// whatever part of the tensor is asked for.  Same generator code as ATen's (hiprand's Philox4x32-10 and its Box-Muller).
#include "common.h"

#include <hiprand/hiprand_kernel.h>

namespace {

__global__ __launch_bounds__(256) void randn_slice_kernel(float* __restrict__ dst, uint64_t seed, uint64_t offset, int64_t threads_total,
                                                          int64_t start, int64_t count) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t li = start + e;
    const int64_t k = li / threads_total;
    const int64_t idx = li - k * threads_total;
    hiprandStatePhilox4_32_10_t state;
    hiprand_init(seed, (unsigned long long)idx, offset + 4ull * (uint64_t)(k >> 2), &state);
    const float4 r = hiprand_normal4(&state);
    const int i = (int)(k & 3);
    dst[e] = i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w;
  }
}

}  // namespace

extern "C" int az_randn_slice_f32(float* dst, uint64_t seed, uint64_t offset, int64_t threads_total, int64_t start, int64_t count,
                                  az_stream_t stream) {
  AZ_REQUIRE(dst, AZ_E_NULL);
  AZ_REQUIRE(threads_total > 0 && threads_total % 256 == 0 && start >= 0 && count > 0 && offset % 4 == 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(randn_slice_kernel, dim3(az_stream_grid(count, 256)), dim3(256), 0, az_s(stream), dst, seed, offset,
                     threads_total, start, count);
  return az_launch_status();
}
