// L2-hit fill bandwidth of one CU's load paths by access shape (measurement only; not part of the library):
//   hipcc --offload-arch=gfx950 -O3 tools/l2_fill_bench.hip -o tools/_build/l2_fill_bench && tools/_build/l2_fill_bench
// Every wave re-reads a small L2-resident region with 16 B per lane, 12 loads in flight, in one of three shapes:
//   rows of 1024 B (a wave instruction = 1 KB contiguous = 8 cache lines), rows of 64 B at a 1536-byte pitch (16 lines), rows of
//   32 B at that pitch (32 lines) -- to VGPRs (buffer_load_dwordx4) or straight to LDS (buffer_load_dwordx4 ... lds).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int ROWB, bool DMA>
__global__ __launch_bounds__(256, 2) void fill(const char* src, float* sink, int iters, unsigned region) {
  __shared__ __attribute__((aligned(1024))) char lds[12 * 1024 * 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, region, 0x00020000);
  constexpr int LPR = ROWB / 16;  // lanes per row
  const unsigned pitch = ROWB == 1024 ? 1024u : 1536u;
  const unsigned voff = (unsigned)((lane / LPR) * pitch + (lane % LPR) * 16);
  const unsigned rows_per_inst = 64 / LPR;
  unsigned base = (blockIdx.x * 4 + wid) * 7919u * 64u % (region / 2);
  base &= ~1023u;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const unsigned so = (base + j * rows_per_inst * pitch) % (region / 2);
      if constexpr (DMA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (wid * 12 + j) * 1024), 16, (int)voff, (int)so, 0, 0);
      else
        v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)so, 0);
    }
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) acc += v[j];
    }
    base = (base + 12 * rows_per_inst * pitch) % (region / 2);
    base &= ~1023u;
  }
  if (acc.x == 0x12345678u) sink[0] = 1.f;
  if (DMA && lds[threadIdx.x] == 77 && iters < 0) sink[1] = 2.f;
}
template <int ROWB, bool DMA>
void run(const char* name, const char* src, float* sink, unsigned region) {
  const int iters = 400, wgs = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<ROWB, DMA>), dim3(wgs), dim3(256), 0, 0, src, sink, iters, region);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * 4 * iters * 12 * 1024;
    if (rep == 2) printf("%-28s %-5s %7.2f TB/s  %6.1f B/clk/CU at 2.4 GHz\n", name, DMA ? "lds" : "vgpr", bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9);
  }
}
int main() {
  const unsigned region = 16u << 20;  // 16 MiB: 2 MiB per XCD if spread evenly -- L2-resident
  char* src; float* sink;
  hipMalloc(&src, region); hipMalloc(&sink, 64);
  hipMemset(src, 1, region);
  run<1024, false>("1 KB contiguous (8 lines)", src, sink, region);
  run<64, false>("64-B rows (16 lines)", src, sink, region);
  run<32, false>("32-B rows (32 lines)", src, sink, region);
  run<1024, true>("1 KB contiguous (8 lines)", src, sink, region);
  run<64, true>("64-B rows (16 lines)", src, sink, region);
  run<32, true>("32-B rows (32 lines)", src, sink, region);
  return 0;
}
