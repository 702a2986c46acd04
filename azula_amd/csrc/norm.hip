// K2 / K6: normalisation kernels on the channel-padded NHWC layout.  All HBM-bound.
//
// GroupNorm is split so that the full-tensor passes are pure streams:
//   stats    : read x once, per-(b, pixel-chunk, group) Chan-combinable partials (n, mean, M2)
//   finalize : fold (mean, rstd, GN affine, AdaZero/FiLM modulation) into per-(b, c) S, T
//   apply    : y = act(x * S + T)           (optionally 2x2 average pooled)
// Reductions: per-thread shifted sums -> (n, mean, M2) -> LDS combine (Chan et al.), so the
// variance never suffers E[x^2] - E[x]^2 cancellation.
//
// Row norms (LayerNorm / RMSNorm over the channel axis, one row = one pixel / token): one
// wave64 per row, wave-shuffle reductions, two-pass variance for parity with torch.var_mean.
#include "common.h"

namespace {

struct Moments {
  float n, mean, m2;
};

__device__ __forceinline__ Moments combine(Moments a, Moments b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  Moments r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean;
  const float f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}

// Fast path: group size Cg % 4 == 0; a slice = `qs` <= 256 float4 channel-chunks holding whole groups, qs | C / 4 (768 and
// 1536 channels -- the ADM decoder's concatenations -- run as 192-chunk slices on 192 of the 256 threads).
// grid = (nchunks, B, slices); slice z covers float4 chunks [z*qs, (z+1)*qs).
// Two-source form: logical channels [0, c0s) live in x (stride c0s), [c0s, cs) in x1 (stride cs - c0s):
// the skip concatenation of the ADM decoder (plugins/adm/_src/unet.py:631) is never materialised.
template <int IO = 0>  // element type of x / x1 (common.h: ld4_io; strides and offsets stay in elements)
__global__ __launch_bounds__(256) void gn_stats_vec_kernel(float* __restrict__ partials,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ x1, int c0s, int64_t HW, int C,
                                                           int cs, int groups, int nchunks, int qs) {
  __shared__ float sh_n[256], sh_mean[256], sh_m2[256];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
  const int Cg = C / groups;
  const int cq = z * qs + (tid % qs);   // float4 chunk index along channels
  const int pl = tid / qs;              // pixel lane
  const int ppi = 256 / qs;             // pixels per iteration (threads >= qs * ppi idle)
  const int64_t ppc = (HW + nchunks - 1) / nchunks;
  const int64_t p0 = (int64_t)chunk * ppc;
  const int64_t p1 = p0 + ppc < HW ? p0 + ppc : HW;
  const int c = cq * 4;
  const bool live = c < C && pl < ppi;
  float s1 = 0.f, s2 = 0.f, cnt = 0.f, shift = 0.f;
  if (live) {
    const bool second = x1 != nullptr && c >= c0s;
    const int scs = x1 == nullptr ? cs : (second ? cs - c0s : c0s);
    const float* src = second ? x1 : x;
    const int64_t base = ((int64_t)b * HW) * scs + (second ? c - c0s : c);  // (element index)
    int64_t p = p0 + pl;
    if (p < p1) shift = ld4_io<IO>(src, base + p * scs).x;  // shift = first element: sums of (x - shift) cannot cancel badly
    // 4 independent 16-byte loads in flight per lane (a streaming reduction is latency-bound otherwise)
    for (; p + 3 * ppi < p1; p += 4 * ppi) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4_io<IO>(src, base + (p + u * ppi) * scs);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a0 = v[u].x - shift, a1 = v[u].y - shift, a2 = v[u].z - shift, a3 = v[u].w - shift;
        s1 += (a0 + a1) + (a2 + a3);
        s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
      cnt += 16.f;
    }
    for (; p < p1; p += ppi) {
      const float4 v = ld4_io<IO>(src, base + p * scs);
      const float a0 = v.x - shift, a1 = v.y - shift, a2 = v.z - shift, a3 = v.w - shift;
      s1 += (a0 + a1) + (a2 + a3);
      s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      cnt += 4.f;
    }
  }
  Moments m;
  m.n = cnt;
  m.mean = cnt > 0.f ? shift + s1 / cnt : 0.f;
  m.m2 = cnt > 0.f ? fmaxf(s2 - s1 * s1 / cnt, 0.f) : 0.f;
  sh_n[tid] = m.n;
  sh_mean[tid] = m.mean;
  sh_m2[tid] = m.m2;
  __syncthreads();
  // two short serial phases instead of one 256-long scan: (1) the first qs threads fold the ppi
  // pixel lanes of their channel chunk, (2) one leader per group folds its Cg/4 chunks.
  if (tid < qs) {
    Moments acc = {sh_n[tid], sh_mean[tid], sh_m2[tid]};
    for (int k = 1; k < ppi; ++k) acc = combine(acc, Moments{sh_n[tid + k * qs], sh_mean[tid + k * qs], sh_m2[tid + k * qs]});
    sh_n[tid] = acc.n;
    sh_mean[tid] = acc.mean;
    sh_m2[tid] = acc.m2;
  }
  __syncthreads();
  const int c_lo = z * qs * 4;
  const int g_lo = c_lo / Cg;
  const int g = g_lo + tid;
  const int c_hi = c_lo + qs * 4 < C ? c_lo + qs * 4 : C;
  if (g < groups && g * Cg < c_hi && (g + 1) * Cg > c_lo) {
    // a group never straddles slices: Cg divides 4 qs (checked on the host)
    const int q_lo = (g * Cg) / 4 - z * qs, q_hi = ((g + 1) * Cg) / 4 - z * qs;
    Moments acc = {0.f, 0.f, 0.f};
    for (int t = q_lo; t < q_hi; ++t) acc = combine(acc, Moments{sh_n[t], sh_mean[t], sh_m2[t]});
    float* out = partials + (((int64_t)b * nchunks + chunk) * groups + g) * 4;
    out[0] = acc.n;
    out[1] = acc.mean;
    out[2] = acc.m2;
    out[3] = 0.f;
  }
}

// Generic path (any C / groups): grid = (nchunks, B, groups), scalar loads.
__global__ __launch_bounds__(256) void gn_stats_generic_kernel(float* __restrict__ partials,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ x1, int c0s, int64_t HW,
                                                               int C, int cs, int groups, int nchunks) {
  __shared__ float sh_n[256], sh_mean[256], sh_m2[256];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y, g = blockIdx.z;
  const int Cg = C / groups;
  const int64_t ppc = (HW + nchunks - 1) / nchunks;
  const int64_t p0 = (int64_t)chunk * ppc;
  const int64_t p1 = p0 + ppc < HW ? p0 + ppc : HW;
  const int64_t total = (p1 > p0 ? p1 - p0 : 0) * Cg;
  float s1 = 0.f, s2 = 0.f, cnt = 0.f, shift = 0.f;
  bool first = true;
  for (int64_t e = tid; e < total; e += 256) {
    const int64_t p = p0 + e / Cg;
    const int c = g * Cg + (int)(e % Cg);
    float v;
    if (x1 != nullptr && c >= c0s) v = x1[((int64_t)b * HW + p) * (cs - c0s) + (c - c0s)];
    else v = x[((int64_t)b * HW + p) * (x1 != nullptr ? c0s : cs) + c];
    if (first) {
      shift = v;
      first = false;
    }
    const float a = v - shift;
    s1 += a;
    s2 += a * a;
    cnt += 1.f;
  }
  sh_n[tid] = cnt;
  sh_mean[tid] = cnt > 0.f ? shift + s1 / cnt : 0.f;
  sh_m2[tid] = cnt > 0.f ? fmaxf(s2 - s1 * s1 / cnt, 0.f) : 0.f;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const Moments r = combine(Moments{sh_n[tid], sh_mean[tid], sh_m2[tid]},
                                Moments{sh_n[tid + o], sh_mean[tid + o], sh_m2[tid + o]});
      sh_n[tid] = r.n;
      sh_mean[tid] = r.mean;
      sh_m2[tid] = r.m2;
    }
    __syncthreads();
  }
  if (tid == 0) {
    float* out = partials + (((int64_t)b * nchunks + chunk) * groups + g) * 4;
    out[0] = sh_n[0];
    out[1] = sh_mean[0];
    out[2] = sh_m2[0];
    out[3] = 0.f;
  }
}

// One wave per (b, group): lanes combine the pixel-chunk partials (Chan), butterfly-merge, then
// write S/T for the group's channels.  Blocks past the last group zero the pad channels.
__global__ __launch_bounds__(64) void gn_finalize_kernel(AzNormFinalizeArgs a) {
  const int b = blockIdx.y;
  const int g = blockIdx.x;
  const int lane = threadIdx.x;
  const int Cg = (int)(a.C / a.groups);
  if (g == a.groups) {  // pad channels [C, cs)
    for (int c = (int)a.C + lane; c < a.cs; c += 64) {
      a.S[(int64_t)b * a.cs + c] = 0.f;
      a.T[(int64_t)b * a.cs + c] = 0.f;
    }
    return;
  }
  // items of this (b, group): partials (n, mean, M2, 0), 16 bytes each.  Folded by two passes of plain sums -- N = sum n,
  // mean = sum(n mean) / N, then M2 = sum(M2_i + n_i (mean_i - mean)^2) -- instead of a chain of pairwise Chan merges with a
  // division each: the launch is pure latency (up to 1024 partials per group on the large maps: 10 - 19 us as a chain).
  // Fixed order, commutative butterfly sums: deterministic, every lane ends with the same bits.
  const float* base;
  int items, qpg = 1, nq = 0, nck = a.nchunks;
  if (a.quads_per_group > 0) {
    // per-(chunk, channel quad) moments written by the producing convolutions (AzConvArgs.gn_quads): group g = quads
    // [g qpg, (g + 1) qpg) of the (possibly two-source) channel axis; a group never straddles the two sources (host check)
    qpg = a.quads_per_group;
    const int q_lo = g * qpg;
    const bool second = q_lo >= a.quads0;
    nq = second ? (int)(a.C / 4) - a.quads0 : a.quads0;
    nck = second && a.nchunks1 > 0 ? a.nchunks1 : a.nchunks;
    base = (second ? a.partials1 : a.partials) + ((int64_t)b * nck * nq + (second ? q_lo - a.quads0 : q_lo)) * 4;
    items = nck * qpg;
  } else {
    base = a.partials + ((int64_t)b * a.nchunks * a.groups + g) * 4;
    nq = a.groups;
    items = a.nchunks;
  }
  // the per-channel factors of the first 64 channels of the group: requested before the reduction, so that their latency
  // overlaps the partials' (the launch is two dependent memory round trips otherwise)
  float w0 = 1.f, bi0 = 0.f, sc0 = 1.f, sh0 = 0.f;
  if (lane < Cg) {
    const int c = g * Cg + lane;
    if (a.weight) w0 = a.weight[c];
    if (a.bias) bi0 = a.bias[c];
    if (a.scale) sc0 = 1.f + a.scale[(int64_t)b * a.scale_bstride + c];
    if (a.shift) sh0 = a.shift[(int64_t)b * a.scale_bstride + c];
  }
  auto item = [&](int i) {  // i -> (chunk k, quad j of the group)
    const int k = i / qpg, j = i - k * qpg;
    return *reinterpret_cast<const float4*>(base + ((int64_t)k * nq + j) * 4);
  };
  float4 keep[4];  // the first four items of a lane stay in registers for the second pass
  float N = 0.f, M1 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = lane + 64 * u;
    keep[u] = i < items ? item(i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    N += keep[u].x;
    M1 += keep[u].x * keep[u].y;
  }
  for (int i = lane + 256; i < items; i += 64) {
    const float4 v = item(i);
    N += v.x;
    M1 += v.x * v.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    N += __shfl_xor(N, o, 64);
    M1 += __shfl_xor(M1, o, 64);
  }
  Moments acc;
  acc.n = N;
  acc.mean = M1 / N;
  float M2 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float d = keep[u].y - acc.mean;
    M2 += keep[u].z + keep[u].x * d * d;
  }
  for (int i = lane + 256; i < items; i += 64) {
    const float4 v = item(i);
    const float d = v.y - acc.mean;
    M2 += v.z + v.x * d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) M2 += __shfl_xor(M2, o, 64);
  acc.m2 = M2;
  const float var = acc.m2 / acc.n;  // biased, as torch.nn.GroupNorm
  const float rstd = rsqrtf(var + a.eps);
  if (lane < Cg) {
    const int c = g * Cg + lane;
    a.S[(int64_t)b * a.cs + c] = rstd * w0 * sc0;
    a.T[(int64_t)b * a.cs + c] = (bi0 - acc.mean * rstd * w0) * sc0 + sh0;
  }
  for (int j = lane + 64; j < Cg; j += 64) {
    const int c = g * Cg + j;
    const float w = a.weight ? a.weight[c] : 1.f;
    const float bi = a.bias ? a.bias[c] : 0.f;
    const float sc = 1.f + (a.scale ? a.scale[(int64_t)b * a.scale_bstride + c] : 0.f);
    const float sh = a.shift ? a.shift[(int64_t)b * a.scale_bstride + c] : 0.f;
    a.S[(int64_t)b * a.cs + c] = rstd * w * sc;
    a.T[(int64_t)b * a.cs + c] = (bi - acc.mean * rstd * w) * sc + sh;
  }
}

template <int IO = 0>
__device__ __forceinline__ float4 ld_cat(const float* __restrict__ x, const float* __restrict__ x1, int c0s, int cs,
                                         int64_t pix, int c) {
  if (x1 == nullptr) return ld4_io<IO>(x, pix * cs + c);
  if (c < c0s) return ld4_io<IO>(x, pix * c0s + c);
  return ld4_io<IO>(x1, pix * (cs - c0s) + (c - c0s));
}

// y = act(x * S[b, c] + T[b, c]).  HBM-bound (8 B per element) only if the per-element ALU work stays small: a
// thread owns ONE channel quad for its whole run (the launch makes the thread count a multiple of the quads per pixel,
// blockIdx.y is the sample), so S / T are loaded once and the loop body has no division -- the first version spent
// three 64-bit divisions per float4 and ran at 2.5 TB/s.
template <int ACT, int IO = 0>  // IO: element type of x / x1 / y
__global__ __launch_bounds__(256) void affine_act_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                         const float* __restrict__ x1, int c0s,
                                                         const float* __restrict__ S, const float* __restrict__ T,
                                                         int HW, int cs) {
  const int q = cs / 4;
  const int b = blockIdx.y;
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int c4 = g % q;
  const int pstride = (gridDim.x * 256) / q;
  const float4 sc = *reinterpret_cast<const float4*>(S + (int64_t)b * cs + c4 * 4);
  const float4 t = *reinterpret_cast<const float4*>(T + (int64_t)b * cs + c4 * 4);
  const int64_t pix0 = (int64_t)b * HW;
  const int64_t yo = (pix0 * q + c4) * 4;  // (element index of the thread's first output quad)
  constexpr int UN = 4;  // loads in flight per thread (2: 0.65 of the HBM roofline at batch 32)
  auto apply = [&](float4 v) {
    float4 o;
    o.x = fmaf(v.x, sc.x, t.x);
    o.y = fmaf(v.y, sc.y, t.y);
    o.z = fmaf(v.z, sc.z, t.z);
    o.w = fmaf(v.w, sc.w, t.w);
    if (ACT == 1) {
      o.x = az_silu(o.x);
      o.y = az_silu(o.y);
      o.z = az_silu(o.z);
      o.w = az_silu(o.w);
    }
    return o;
  };
  int p = g / q;
  for (; p + (UN - 1) * pstride < HW; p += UN * pstride) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = ld_cat<IO>(x, x1, c0s, cs, pix0 + p + u * pstride, c4 * 4);
#pragma unroll
    for (int u = 0; u < UN; ++u) st4_io<IO>(y, yo + (int64_t)(p + u * pstride) * cs, apply(v[u]));
  }
  for (; p < HW; p += pstride) st4_io<IO>(y, yo + (int64_t)p * cs, apply(ld_cat<IO>(x, x1, c0s, cs, pix0 + p, c4 * 4)));
}

template <int ACT, int PH, int IO = 0>  // PH = rows of the pooling window: 2 (AvgPool2d(2, 2)) or 1 (AvgPool1d(2) on a one-row image)
__global__ __launch_bounds__(256) void affine_act_pool_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                              const float* __restrict__ x1, int c0s,
                                                              const float* __restrict__ S,
                                                              const float* __restrict__ T, int64_t B, int H, int W,
                                                              int cs) {
  const int q = cs / 4;
  const int Ho = H / PH, Wo = W / 2;
  const int64_t per_b = (int64_t)Ho * Wo * q;
  const int64_t total = B * per_b;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / per_b;
    const int64_t r = e - b * per_b;
    const int c4 = (int)(r % q);
    const int ow = (int)((r / q) % Wo);
    const int oh = (int)(r / ((int64_t)q * Wo));
    const float4 s = *reinterpret_cast<const float4*>(S + b * cs + c4 * 4);
    const float4 t = *reinterpret_cast<const float4*>(T + b * cs + c4 * 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < PH; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int64_t pix = (b * H + (PH * oh + dy)) * W + (2 * ow + dx);
        const float4 v = ld_cat<IO>(x, x1, c0s, cs, pix, c4 * 4);
        float4 o;
        o.x = fmaf(v.x, s.x, t.x);
        o.y = fmaf(v.y, s.y, t.y);
        o.z = fmaf(v.z, s.z, t.z);
        o.w = fmaf(v.w, s.w, t.w);
        if (ACT == 1) {
          o.x = az_silu(o.x);
          o.y = az_silu(o.y);
          o.z = az_silu(o.z);
          o.w = az_silu(o.w);
        }
        acc.x += o.x;
        acc.y += o.y;
        acc.z += o.z;
        acc.w += o.w;
      }
    constexpr float inv = PH == 2 ? 0.25f : 0.5f;
    acc.x *= inv;
    acc.y *= inv;
    acc.z *= inv;
    acc.w *= inv;
    st4_io<IO>(y, e * 4, acc);
  }
}

// rownorm_mod_kernel's fast path on 2-byte rows (IO 1: bfloat16, 2: IEEE half): one wave per row, lane -> 8 consecutive values per
// 16-byte load (c = 8 lane + 512 k), fp32 statistics in the fast path's order of operations, output rounded to nearest even.
template <int IO>
__global__ __launch_bounds__(256) void rownorm_mod_h16_kernel(unsigned short* __restrict__ y, const unsigned short* __restrict__ x,
                                                              const float* __restrict__ weight, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int64_t scale_bstride, int64_t rows,
                                                              int64_t rows_per_batch, int C, int kind, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  constexpr int NV = 8;
  const int c0 = lane * 8;
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float* xr = reinterpret_cast<const float*>(x + row * C);  // (typed accesses through ld4_io: element indices)
    float* yr = reinterpret_cast<float*>(y + row * C);
    const int64_t b = row / rows_per_batch;
    float4 v[NV][2];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = c0 + 512 * k;
      if (c < C) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          v[k][hh] = ld4_io<IO>(xr, c + 4 * hh);
          const float4 t = v[k][hh];
          if (kind == 0) s += (t.x + t.y) + (t.z + t.w);
          else s += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        }
      }
    }
    s = az_wave_sum(s);
    float mean = 0.f, rstd;
    if (kind == 0) {
      mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (c0 + 512 * k < C) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const float a0 = v[k][hh].x - mean, a1 = v[k][hh].y - mean, a2 = v[k][hh].z - mean, a3 = v[k][hh].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
          }
        }
      q = az_wave_sum(q);
      rstd = rsqrtf(q / (float)(C - 1) + eps);
    } else {
      rstd = rsqrtf(s / (float)C + eps);
    }
    const float* sc = scale ? scale + b * scale_bstride : nullptr;
    const float* sh = shift ? shift + b * scale_bstride : nullptr;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cb = c0 + 512 * k;
      if (cb < C) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int c = cb + 4 * hh;
          const float in[4] = {v[k][hh].x, v[k][hh].y, v[k][hh].z, v[k][hh].w};
          float w4[4] = {1.f, 1.f, 1.f, 1.f}, sc4[4] = {0.f, 0.f, 0.f, 0.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
          if (weight) { const float4 t = *reinterpret_cast<const float4*>(weight + c); w4[0] = t.x; w4[1] = t.y; w4[2] = t.z; w4[3] = t.w; }
          if (sc) { const float4 t = *reinterpret_cast<const float4*>(sc + c); sc4[0] = t.x; sc4[1] = t.y; sc4[2] = t.z; sc4[3] = t.w; }
          if (sh) { const float4 t = *reinterpret_cast<const float4*>(sh + c); sh4[0] = t.x; sh4[1] = t.y; sh4[2] = t.z; sh4[3] = t.w; }
          float o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float o = (in[j] - mean) * rstd;
            if (weight) o *= w4[j];
            o4[j] = o * (1.f + sc4[j]) + sh4[j];
          }
          st4_io<IO>(yr, c, make_float4(o4[0], o4[1], o4[2], o4[3]));
        }
      }
    }
  }
}

// One wave per row.  kind 0: layer norm (unbiased variance), kind 1: RMS norm.
__global__ __launch_bounds__(256) void rownorm_mod_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                          const float* __restrict__ weight,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int64_t scale_bstride,
                                                          int64_t rows, int64_t rows_per_batch, int C, int cs,
                                                          int kind, float eps, int fast) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const bool vec = (C % 4) == 0;
  // Fast path (token widths of the ViT / DiT / JiT backbones): the row lives in registers (<= 8 float4 per lane), is
  // read ONCE with 16-byte loads and written with 16-byte stores; the summation order is that of the generic path
  // below (lane-sequential over c = 4 lane + 256 k, then the wave butterfly), so the results are bit-identical.
  constexpr int NV = 8;
  if (fast) {  // host: C % 4 == 0, C == cs, C <= 2048, weight / scale / shift 16-byte aligned rows
    const int c0 = lane * 4;
    for (int64_t row = wave; row < rows; row += nwaves) {
      const float* xr = x + row * cs;
      float* yr = y + row * cs;
      const int64_t b = row / rows_per_batch;
      float4 v[NV];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = c0 + 256 * k;
        if (c < C) {
          v[k] = *reinterpret_cast<const float4*>(xr + c);
          if (kind == 0) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
          else s += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
      }
      s = az_wave_sum(s);
      float mean = 0.f, rstd;
      if (kind == 0) {
        mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
          if (c0 + 256 * k < C) {
            const float a0 = v[k].x - mean, a1 = v[k].y - mean, a2 = v[k].z - mean, a3 = v[k].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
          }
        q = az_wave_sum(q);
        rstd = rsqrtf(q / (float)(C - 1) + eps);
      } else {
        rstd = rsqrtf(s / (float)C + eps);
      }
      const float* sc = scale ? scale + b * scale_bstride : nullptr;
      const float* sh = shift ? shift + b * scale_bstride : nullptr;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = c0 + 256 * k;
        if (c < C) {
          const float in[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
          float w4[4] = {1.f, 1.f, 1.f, 1.f}, sc4[4] = {0.f, 0.f, 0.f, 0.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
          if (weight) { const float4 t = *reinterpret_cast<const float4*>(weight + c); w4[0] = t.x; w4[1] = t.y; w4[2] = t.z; w4[3] = t.w; }
          if (sc) { const float4 t = *reinterpret_cast<const float4*>(sc + c); sc4[0] = t.x; sc4[1] = t.y; sc4[2] = t.z; sc4[3] = t.w; }
          if (sh) { const float4 t = *reinterpret_cast<const float4*>(sh + c); sh4[0] = t.x; sh4[1] = t.y; sh4[2] = t.z; sh4[3] = t.w; }
          float o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float o = (in[j] - mean) * rstd;
            if (weight) o *= w4[j];
            o4[j] = o * (1.f + sc4[j]) + sh4[j];
          }
          *reinterpret_cast<float4*>(yr + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
      }
    }
    return;
  }
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float* xr = x + row * cs;
    float* yr = y + row * cs;
    const int64_t b = row / rows_per_batch;
    float s = 0.f;
    if (vec) {
      for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        if (kind == 0) s += (v.x + v.y) + (v.z + v.w);
        else s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float v = xr[c];
        s += kind == 0 ? v : v * v;
      }
    }
    s = az_wave_sum(s);
    float mean = 0.f, rstd;
    if (kind == 0) {
      mean = s / (float)C;
      float q = 0.f;
      if (vec) {
        for (int c = lane * 4; c < C; c += 256) {
          const float4 v = *reinterpret_cast<const float4*>(xr + c);
          const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
          q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
      } else {
        for (int c = lane; c < C; c += 64) {
          const float a0 = xr[c] - mean;
          q += a0 * a0;
        }
      }
      q = az_wave_sum(q);
      rstd = rsqrtf(q / (float)(C - 1) + eps);
    } else {
      rstd = rsqrtf(s / (float)C + eps);
    }
    const float* sc = scale ? scale + b * scale_bstride : nullptr;
    const float* sh = shift ? shift + b * scale_bstride : nullptr;
    for (int c = lane; c < cs; c += 64) {
      float o = 0.f;
      if (c < C) {
        o = (xr[c] - mean) * rstd;
        if (weight) o *= weight[c];
        o = o * (1.f + (sc ? sc[c] : 0.f)) + (sh ? sh[c] : 0.f);
      }
      yr[c] = o;
    }
  }
}

// slots[block] = max |x| over the float4s the block visits (grid-stride, four loads in flight per thread); every slot is written.
__global__ __launch_bounds__(256) void absmax_kernel(float* __restrict__ slots, const float* __restrict__ x, int64_t n4, int64_t n) {
  __shared__ float sh[4];
  const int64_t stride = (int64_t)gridDim.x * 256;
  float m = 0.f;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + 4 * (i + u * stride));
#pragma unroll
    for (int u = 0; u < 4; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
  }
  for (; i < n4; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(x + 4 * i);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));  // (a tail of < 4 elements)
  m = az_wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) slots[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// The same slots from GroupNorm partial moments (n, mean, M2, -) that a producing convolution left (AzConvArgs.gn_quads): every element
// of a partial satisfies |x| <= |mean| + sqrt(M2), so the largest such bound is an upper bound of max |x| -- loose by up to sqrt(n),
// which costs an f16x2 launch a few of its 30 binades of headroom and no precision -- for the price of reading the partials.
__global__ __launch_bounds__(256) void absmax_moments_kernel(float* __restrict__ slots, const float4* __restrict__ partials, int64_t count) {
  __shared__ float sh[4];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const float4 q = partials[i];
    if (q.x > 0.f) m = fmaxf(m, fabsf(q.y) + sqrtf(fmaxf(q.z, 0.f)));
  }
  m = az_wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) slots[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

}  // namespace

template <int IO>
static int affine_act_h16_launch(float* y, const float* x, const float* x1, int64_t c0s, const float* S, const float* T, int64_t B,
                                 int64_t H, int64_t W, int64_t cs, int32_t act, int32_t pool, hipStream_t st) {
  if (pool == 2) {
    AZ_REQUIRE(W % 2 == 0, AZ_E_SHAPE);
    const int grid = az_stream_grid(B * H * (W / 2) * (cs / 4), 256);
    if (act == 1) hipLaunchKernelGGL((affine_act_pool_kernel<1, 1, IO>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H, (int)W, (int)cs);
    else hipLaunchKernelGGL((affine_act_pool_kernel<0, 1, IO>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H, (int)W, (int)cs);
  } else if (pool) {
    AZ_REQUIRE(H % 2 == 0 && W % 2 == 0, AZ_E_SHAPE);
    const int grid = az_stream_grid(B * (H / 2) * (W / 2) * (cs / 4), 256);
    if (act == 1) hipLaunchKernelGGL((affine_act_pool_kernel<1, 2, IO>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H, (int)W, (int)cs);
    else hipLaunchKernelGGL((affine_act_pool_kernel<0, 2, IO>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H, (int)W, (int)cs);
  } else {
    AZ_REQUIRE(H * W < (1ll << 31) && B < 65536, AZ_E_SHAPE);
    const int q = (int)(cs / 4);
    int qq = q, r256 = 256;
    while (r256) { const int tmp = qq % r256; qq = r256; r256 = tmp; }  // qq = gcd(q, 256)
    const int unit = q / qq;
    int64_t want = (H * W * q + 256 * 8 - 1) / (256 * 8);
    const int64_t cap = (2048 + B - 1) / B;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    const int gx = (int)((want + unit - 1) / unit * unit);
    dim3 grid((unsigned)gx, (unsigned)B);
    if (act == 1) hipLaunchKernelGGL((affine_act_kernel<1, IO>), grid, dim3(256), 0, st, y, x, x1, (int)c0s, S, T, (int)(H * W), (int)cs);
    else hipLaunchKernelGGL((affine_act_kernel<0, IO>), grid, dim3(256), 0, st, y, x, x1, (int)c0s, S, T, (int)(H * W), (int)cs);
  }
  return az_launch_status();
}

extern "C" {

int az_absmax_f32(float* slots, const float* x, int64_t n, az_stream_t stream) {
  AZ_REQUIRE(slots && x, AZ_E_NULL);
  AZ_REQUIRE(n > 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(x) && AZ_ALIGNED16(slots), AZ_E_ALIGN);
  hipLaunchKernelGGL(absmax_kernel, dim3(AZ_ABSMAX_SLOTS), dim3(256), 0, az_s(stream), slots, x, n / 4, n);
  return az_launch_status();
}

int az_absmax_from_moments_f32(float* slots, const float* partials, int64_t count, az_stream_t stream) {
  AZ_REQUIRE(slots && partials, AZ_E_NULL);
  AZ_REQUIRE(count > 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(partials) && AZ_ALIGNED16(slots), AZ_E_ALIGN);
  hipLaunchKernelGGL(absmax_moments_kernel, dim3(AZ_ABSMAX_SLOTS), dim3(256), 0, az_s(stream), slots, reinterpret_cast<const float4*>(partials), count);
  return az_launch_status();
}

int az_groupnorm_stats_f32(float* partials, const float* x, const float* x1, int64_t c0s, int64_t B, int64_t HW,
                           int64_t C, int64_t cs, int32_t groups, int32_t nchunks, az_stream_t stream) {
  AZ_REQUIRE(partials && x, AZ_E_NULL);
  if (x1) AZ_REQUIRE(c0s > 0 && c0s < cs && c0s % 4 == 0 && C == cs && AZ_ALIGNED16(x1), AZ_E_SHAPE);
  AZ_REQUIRE(B > 0 && HW > 0 && C > 0 && cs >= C && cs % 4 == 0 && groups > 0 && C % groups == 0 && nchunks > 0,
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(x), AZ_E_ALIGN);
  const int Cg = (int)(C / groups);
  const int q = (int)(cs / 4);
  int qs = 0;  // float4 chunks per slice: the largest divisor of q, at most 256, that holds whole groups
  if (Cg % 4 == 0 && C == cs) {
    for (int d = q < 256 ? q : 256; d >= Cg / 4 && qs == 0; --d)
      if (q % d == 0 && (4 * d) % Cg == 0) qs = d;
  } else if (Cg % 4 == 0 && q <= 256) {
    qs = q;  // (padded channel stride: one slice, the pad chunk is not live)
  }
  const bool fast = qs >= 32 || (qs > 0 && qs == q);  // (narrow slices of a wide tensor: the generic kernel)
  if (fast) {
    dim3 grid((unsigned)nchunks, (unsigned)B, (unsigned)(q / qs));
    hipLaunchKernelGGL(gn_stats_vec_kernel<0>, grid, dim3(256), 0, az_s(stream), partials, x, x1, (int)c0s, HW, (int)C,
                       (int)cs, (int)groups, (int)nchunks, qs);
  } else {
    dim3 grid((unsigned)nchunks, (unsigned)B, (unsigned)groups);
    hipLaunchKernelGGL(gn_stats_generic_kernel, grid, dim3(256), 0, az_s(stream), partials, x, x1, (int)c0s, HW, (int)C,
                       (int)cs, (int)groups, (int)nchunks);
  }
  return az_launch_status();
}

int az_groupnorm_finalize_f32(const AzNormFinalizeArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->S && a->T && a->partials, AZ_E_NULL);
  AZ_REQUIRE(a->B > 0 && a->C > 0 && a->cs >= a->C && a->groups > 0 && a->C % a->groups == 0 && a->nchunks > 0 &&
                 a->nchunks1 >= 0,
             AZ_E_SHAPE);
  if (a->quads_per_group > 0) {  // partials from conv epilogues: whole quads per group, groups inside one source
    const int64_t Cg = a->C / a->groups;
    AZ_REQUIRE(a->C % 4 == 0 && Cg == 4ll * a->quads_per_group && a->quads0 > 0 && a->quads0 <= a->C / 4 &&
                   a->quads0 % a->quads_per_group == 0 && (a->quads0 == a->C / 4 || a->partials1 != nullptr),
               AZ_E_SHAPE);
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)a->groups + 1, (unsigned)a->B), dim3(64), 0, az_s(stream), *a);
  return az_launch_status();
}

int az_affine_act_f32(float* y, const float* x, const float* x1, int64_t c0s, const float* S, const float* T,
                      int64_t B, int64_t H, int64_t W, int64_t cs, int32_t act, int32_t pool, az_stream_t stream) {
  AZ_REQUIRE(y && x && S && T, AZ_E_NULL);
  if (x1) AZ_REQUIRE(c0s > 0 && c0s < cs && c0s % 4 == 0 && AZ_ALIGNED16(x1), AZ_E_SHAPE);
  AZ_REQUIRE(B > 0 && H > 0 && W > 0 && cs > 0 && cs % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(y) && AZ_ALIGNED16(x) && AZ_ALIGNED16(S) && AZ_ALIGNED16(T), AZ_E_ALIGN);
  hipStream_t st = az_s(stream);
  if (pool == 2) {  // width only: AvgPool1d(2) of a (B, C, L) signal held as a one-row image
    AZ_REQUIRE(W % 2 == 0, AZ_E_SHAPE);
    const int grid = az_stream_grid(B * H * (W / 2) * (cs / 4), 256);
    if (act == 1)
      hipLaunchKernelGGL((affine_act_pool_kernel<1, 1>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H, (int)W, (int)cs);
    else
      hipLaunchKernelGGL((affine_act_pool_kernel<0, 1>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H, (int)W, (int)cs);
  } else if (pool) {
    AZ_REQUIRE(H % 2 == 0 && W % 2 == 0, AZ_E_SHAPE);
    const int grid = az_stream_grid(B * (H / 2) * (W / 2) * (cs / 4), 256);
    if (act == 1)
      hipLaunchKernelGGL((affine_act_pool_kernel<1, 2>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H,
                         (int)W, (int)cs);
    else
      hipLaunchKernelGGL((affine_act_pool_kernel<0, 2>), dim3(grid), dim3(256), 0, st, y, x, x1, (int)c0s, S, T, B, (int)H,
                         (int)W, (int)cs);
  } else {
    AZ_REQUIRE(H * W < (1ll << 31) && B < 65536, AZ_E_SHAPE);
    // threads per sample: a multiple of the quads per pixel q (so a thread keeps its channel quad), about 8
    // float4 per thread, at most ~2048 workgroups over the batch
    const int q = (int)(cs / 4);
    int qq = q, r256 = 256;
    while (r256) { const int tmp = qq % r256; qq = r256; r256 = tmp; }  // qq = gcd(q, 256)
    const int unit = q / qq;  // workgroups per sample must be a multiple of this
    int64_t want = (H * W * q + 256 * 8 - 1) / (256 * 8);
    const int64_t cap = (2048 + B - 1) / B;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    const int gx = (int)((want + unit - 1) / unit * unit);
    dim3 grid((unsigned)gx, (unsigned)B);
    if (act == 1)
      hipLaunchKernelGGL((affine_act_kernel<1, 0>), grid, dim3(256), 0, st, y, x, x1, (int)c0s, S, T, (int)(H * W), (int)cs);
    else
      hipLaunchKernelGGL((affine_act_kernel<0, 0>), grid, dim3(256), 0, st, y, x, x1, (int)c0s, S, T, (int)(H * W), (int)cs);
  }
  return az_launch_status();
}

/* az_groupnorm_stats_f32 / az_affine_act_f32 on tensors held in a 2-byte type (dtype 1: bfloat16, 2: IEEE half; x, x1 and y alike):
 * the activations of a module cast to half precision.  Statistics and the apply arithmetic are fp32; y is rounded to nearest even.
 * Channel strides are multiples of 8; the statistics need whole groups in 4-channel chunks (else AZ_E_UNSUPPORTED).            */
int az_groupnorm_stats_h16(float* partials, const void* x, const void* x1, int64_t c0s, int64_t B, int64_t HW, int64_t C,
                           int64_t cs, int32_t groups, int32_t nchunks, int32_t dtype, az_stream_t stream) {
  AZ_REQUIRE(partials && x, AZ_E_NULL);
  AZ_REQUIRE(dtype == 1 || dtype == 2, AZ_E_SHAPE);
  if (x1) AZ_REQUIRE(c0s > 0 && c0s < cs && c0s % 8 == 0 && C == cs && AZ_ALIGNED16(x1), AZ_E_SHAPE);
  AZ_REQUIRE(B > 0 && HW > 0 && C > 0 && cs >= C && cs % 8 == 0 && groups > 0 && C % groups == 0 && nchunks > 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(x), AZ_E_ALIGN);
  const int Cg = (int)(C / groups);
  const int q = (int)(cs / 4);
  int qs = 0;
  if (Cg % 4 == 0 && C == cs) {
    for (int d = q < 256 ? q : 256; d >= Cg / 4 && qs == 0; --d)
      if (q % d == 0 && (4 * d) % Cg == 0) qs = d;
  } else if (Cg % 4 == 0 && q <= 256) {
    qs = q;
  }
  AZ_REQUIRE(qs > 0, AZ_E_UNSUPPORTED);  // (the scalar generic kernel has no typed form)
  dim3 grid((unsigned)nchunks, (unsigned)B, (unsigned)(q / qs));
  const float* xf = reinterpret_cast<const float*>(x);
  const float* x1f = reinterpret_cast<const float*>(x1);
  if (dtype == 1)
    hipLaunchKernelGGL(gn_stats_vec_kernel<1>, grid, dim3(256), 0, az_s(stream), partials, xf, x1f, (int)c0s, HW, (int)C, (int)cs, (int)groups, (int)nchunks, qs);
  else
    hipLaunchKernelGGL(gn_stats_vec_kernel<2>, grid, dim3(256), 0, az_s(stream), partials, xf, x1f, (int)c0s, HW, (int)C, (int)cs, (int)groups, (int)nchunks, qs);
  return az_launch_status();
}

int az_affine_act_h16(void* y, const void* x, const void* x1, int64_t c0s, const float* S, const float* T, int64_t B, int64_t H,
                      int64_t W, int64_t cs, int32_t act, int32_t pool, int32_t dtype, az_stream_t stream) {
  AZ_REQUIRE(y && x && S && T, AZ_E_NULL);
  AZ_REQUIRE(dtype == 1 || dtype == 2, AZ_E_SHAPE);
  if (x1) AZ_REQUIRE(c0s > 0 && c0s < cs && c0s % 8 == 0 && AZ_ALIGNED16(x1), AZ_E_SHAPE);
  AZ_REQUIRE(B > 0 && H > 0 && W > 0 && cs > 0 && cs % 8 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(y) && AZ_ALIGNED16(x) && AZ_ALIGNED16(S) && AZ_ALIGNED16(T), AZ_E_ALIGN);
  float* yf = reinterpret_cast<float*>(y);
  const float* xf = reinterpret_cast<const float*>(x);
  const float* x1f = reinterpret_cast<const float*>(x1);
  if (dtype == 1) return affine_act_h16_launch<1>(yf, xf, x1f, c0s, S, T, B, H, W, cs, act, pool, az_s(stream));
  return affine_act_h16_launch<2>(yf, xf, x1f, c0s, S, T, B, H, W, cs, act, pool, az_s(stream));
}

/* az_rownorm_mod_f32 on rows held in a 2-byte type (dtype 1: bfloat16, 2: IEEE half; x and y alike): the activations of a module
 * cast to half precision.  Statistics, the gain and the modulation are fp32 (the reference's RMSNorm / LayerNorm upcast
 * internally); the row is read once into registers (C % 8 == 0, C == cs, C <= 4096).                                          */
int az_rownorm_mod_h16(void* y, const void* x, const float* weight, const float* scale, const float* shift,
                       int64_t scale_bstride, int64_t rows, int64_t rows_per_batch, int64_t C, int64_t cs, int32_t kind,
                       float eps, int32_t dtype, az_stream_t stream) {
  AZ_REQUIRE(y && x, AZ_E_NULL);
  AZ_REQUIRE(rows > 0 && rows_per_batch > 0 && C > 0 && (kind == 0 || kind == 1) && (dtype == 1 || dtype == 2), AZ_E_SHAPE);
  AZ_REQUIRE(C % 8 == 0 && C == cs && C <= 4096 && scale_bstride % 4 == 0, AZ_E_UNSUPPORTED);
  AZ_REQUIRE(AZ_ALIGNED16(y) && AZ_ALIGNED16(x) && AZ_ALIGNED16(weight) && AZ_ALIGNED16(scale) && AZ_ALIGNED16(shift), AZ_E_ALIGN);
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (dtype == 1)
    hipLaunchKernelGGL(rownorm_mod_h16_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, az_s(stream), (unsigned short*)y, (const unsigned short*)x,
                       weight, scale, shift, scale_bstride, rows, rows_per_batch, (int)C, (int)kind, eps);
  else
    hipLaunchKernelGGL(rownorm_mod_h16_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, az_s(stream), (unsigned short*)y, (const unsigned short*)x,
                       weight, scale, shift, scale_bstride, rows, rows_per_batch, (int)C, (int)kind, eps);
  return az_launch_status();
}

int az_rownorm_mod_f32(float* y, const float* x, const float* weight, const float* scale, const float* shift,
                       int64_t scale_bstride,
                       int64_t rows, int64_t rows_per_batch, int64_t C, int64_t cs, int32_t kind, float eps,
                       az_stream_t stream) {
  AZ_REQUIRE(y && x, AZ_E_NULL);
  AZ_REQUIRE(rows > 0 && rows_per_batch > 0 && C > 0 && cs >= C && cs % 4 == 0 && (kind == 0 || kind == 1),
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(y) && AZ_ALIGNED16(x), AZ_E_ALIGN);
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  const int fast = C % 4 == 0 && C == cs && C <= 2048 && AZ_ALIGNED16(weight) && AZ_ALIGNED16(scale) &&
                   AZ_ALIGNED16(shift) && scale_bstride % 4 == 0;
  hipLaunchKernelGGL(rownorm_mod_kernel, dim3((unsigned)blocks), dim3(256), 0, az_s(stream), y, x, weight, scale,
                     shift, scale_bstride, rows, rows_per_batch, (int)C, (int)cs, (int)kind, eps, fast);
  return az_launch_status();
}

}  // extern "C"
