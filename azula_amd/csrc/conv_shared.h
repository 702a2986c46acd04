// Shared device helpers of the convolution kernels (conv.hip, wino_x3.hip): bounds-checked buffer loads, the depth-tap plane
// arithmetic and the fused epilogue (bias, activation, gate, residual, split-K slabs, GroupNorm moments, q/k preparation).
#pragma once

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// (ld4_io / st4_io / st2_io / round_io -- typed 4-element accesses of half-precision activation tensors -- live in common.h)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer load: per-lane byte offset `voff` (bounds-checked: >= num_records returns 0,
// which is how padding, ragged tiles and channel tails are zero-filled without branches) plus a
// wave-uniform byte offset `soff` (not bounds-checked) that walks the K dimension.
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(float4, v);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 buf_ld2(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(f32x2, v);
}

// Per-lane offset of a tap that must read as zero (padding, ragged tiles, channel tails).  Every descriptor's
// num_records is clamped to <= OOB (the `clamp_bytes` lambdas), so this offset is out of range for ANY tensor size; the
// legitimate offsets of a workgroup are relative to its first sample and validated on the host to stay below 2^31.
// (Round 1 clamped at 0xFFFFFFF0: with more than 2 GiB of activations behind a workgroup's first sample -- ADM's
// 512-channel concatenation at 256^2, batch 32 -- the "zero" taps of the first 16 samples read sample b + 16 instead.)
constexpr unsigned OOB = 0x80000000u;

// Circular ("periodic") padding -- azula/nn/unet.py:175-180, torch padding_mode="circular": an out-of-range tap
// coordinate wraps around the map instead of reading zero.  Only in the per-(tap, source) address setup, never in a K loop.
__device__ __forceinline__ int wrap_coord(int i, int n, int mode) {
  if (mode) {
    i %= n;
    if (i < 0) i += n;
  }
  return i;
}
#define AZ_RSRC_CLAMP 0x80000000ll  // == OOB: the largest num_records a descriptor may carry

// AzConvArgs.depth (one depth tap of a 3-D convolution over all planes of all volumes): the source plane of image b is
// b + depth_shift inside b's own volume -- outside it the tap reads zeros (`ok` false) or, with depth_wrap (circular padding
// along the depth axis), the plane at the other end of the volume.
__device__ __forceinline__ int az_depth_plane(const AzConvArgs& a, int b, bool& ok) {
  ok = true;
  if (a.depth <= 0) return b;
  const int bd = b % a.depth;
  int sb = bd + a.depth_shift;
  if (a.depth_wrap) sb = sb < 0 ? sb + a.depth : (sb >= a.depth ? sb - a.depth : sb);
  else ok = (unsigned)sb < (unsigned)a.depth;
  return b - bd + sb;
}
// First plane the descriptors of a tile start at (they then run to the last plane, inside the allocation whatever the shift):
// zero padding: the first image's source plane (clamped at 0); circular: the start of the first image's volume.
__device__ __forceinline__ int az_depth_base(const AzConvArgs& a, int b_first) {
  if (a.depth <= 0) return b_first;
  if (a.depth_wrap) return b_first - b_first % a.depth;
  const int s = b_first + a.depth_shift;
  return s > 0 ? s : 0;
}

// log2 upsampling factor along the width: its own field when the descriptor is anisotropic (validated to [0, 4])
static inline int az_upw(const AzConvArgs* a, int up, int up_w) { return a->aniso ? up_w : up; }  // (range checked by the callers)

// Epilogue for 4 consecutive output channels [co, co+4) of output pixel n.
// The fused epilogue in two halves so that callers can issue the loads of several outputs before the first store (the
// compiler cannot move a load above a store that might alias it): fetch = the gate / residual reads, apply = bias,
// activation, gate, residual, store.  The arithmetic order is fixed: ((v + bias) -> act) * gate + res (act 6: silu(v + bias + res)).
__device__ __forceinline__ int64_t epilogue_res_index(const AzConvArgs& a, int n, int b) {
  const int rem = n - b * (a.hout * a.wout);
  if (a.res_up) {
    const int oh = rem / a.wout, ow = rem - oh * a.wout;
    return ((int64_t)b * a.hres + (oh >> 1)) * a.wres + (ow >> 1);
  }
  return a.res_bcast ? rem : n;
}

template <int IO = 0>
__device__ __forceinline__ void epilogue_fetch(const AzConvArgs& a, int n, int b, int co, float4& gate, float4& res) {
  if (a.gate) gate = ld4(a.gate + (int64_t)b * a.gate_bstride + co);
  if (a.res) res = ld4_io<IO>(a.res, epilogue_res_index(a, n, b) * a.cout_s + co);
}

template <int IO = 0>  // (the planar destination -- the network's output -- is always fp32)
__device__ __forceinline__ float4 epilogue_apply_store(const AzConvArgs& a, int n, int b, int co, float4 v, float4 bv,
                                                       float4 g, float4 r) {  // bv: bias (zeros if none); returns what it stored
  if (a.bias) {
    v.x += bv.x;
    v.y += bv.y;
    v.z += bv.z;
    v.w += bv.w;
  }
  if (a.act == 4) {
    // SwiGLU over interleaved channel pairs, y[c] = x[2c] * silu(x[2c+1]) (azula/nn/layers.py:107-110; JiT's SwiGLUFFN with
    // its w12 rows interleaved at build time): the output has HALF the channels (row stride cout_s / 2; the host admits no
    // gate / residual / planar destination here) -- the separate az_swiglu_f32 pass and its 12 B per pair are gone
    st2_io<IO>(a.dst, (int64_t)n * (a.cout_s / 2) + co / 2, make_float2(v.x * az_silu(v.y), v.z * az_silu(v.w)));
    return v;
  }
  if (a.act == 6) {
    // SiLU of the SUM with the residual operand (no gate, NHWC destination): the last depth tap of a 3-D convolution that is
    // followed by an activation accumulates into the other taps' sum and activates it in the same store (nn/unet3d.py)
    v = round_io<IO>(make_float4(az_silu(v.x + r.x), az_silu(v.y + r.y), az_silu(v.z + r.z), az_silu(v.w + r.w)));
    st4_io<IO>(a.dst, (int64_t)n * a.cout_s + co, v);
    return v;
  }
  if (a.act == 1) {
    v.x = az_silu(v.x);
    v.y = az_silu(v.y);
    v.z = az_silu(v.z);
    v.w = az_silu(v.w);
  } else if (a.act >= 2) {  // 2: ReLU, 3: ReLU^2 (azula/nn/layers.py:85-86)
    v.x = fmaxf(v.x, 0.f);
    v.y = fmaxf(v.y, 0.f);
    v.z = fmaxf(v.z, 0.f);
    v.w = fmaxf(v.w, 0.f);
    if (a.act == 3) {
      v.x *= v.x;
      v.y *= v.y;
      v.z *= v.z;
      v.w *= v.w;
    }
  }
  if (a.gate) {
    v.x *= g.x;
    v.y *= g.y;
    v.z *= g.z;
    v.w *= g.w;
  }
  if (a.res) {
    v.x += r.x;
    v.y += r.y;
    v.z += r.z;
    v.w += r.w;
  }
  if (a.dst_nchw) {
    const int hw = a.hout * a.wout;
    const int rem = n - b * hw;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (co + j < a.dst_c) a.dst[((int64_t)b * a.dst_c + co + j) * hw + rem] = vv[j];
  } else {
    v = round_io<IO>(v);
    st4_io<IO>(a.dst, (int64_t)n * a.cout_s + co, v);
  }
  return v;
}

template <int IO = 0>
__device__ __forceinline__ void epilogue_store_b(const AzConvArgs& a, int n, int b, int co, float4 v) {  // b = image of pixel n
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f), r = g, bv = g;
  if (a.bias) bv = ld4(a.bias + co);
  epilogue_fetch<IO>(a, n, b, co, g, r);
  epilogue_apply_store<IO>(a, n, b, co, v, bv, g, r);
}

// A batch of NB outputs of one thread (same channel quad `co`, pixels n[i] of images b[i]; n[i] < 0: skip): all gate /
// residual reads are issued first, then the NB stores.
// Straight-line form of the batch for the NHWC destination with act in {none, SiLU}: (ACT, GATE, RES) are compile-time,
// so the NB iterations contain no scalar branch -- only exec-masked loads / stores.  With branches in the body the
// compiler cannot count outstanding memory operations across them and waits with vmcnt(0) at the top of every
// iteration, i.e. for the previous iteration's STORE to be acknowledged (stores count in vmcnt on gfx9): the store
// phase of a workgroup was 8 serialised L2 round trips.  RES: 0 none, 1 same pixel, 2 upsampled / broadcast index.
// MOM: also accumulate the moments of the STORED values about the first one (mom = {pivot, sum d, sum d^2}) on the fly
// (nothing but three registers outlives the stores).
template <int NB, bool MOM, int ACT, bool GATE, int RES, int IO = 0>
__device__ __forceinline__ void epilogue_batch_nhwc(const AzConvArgs& a, const int (&n)[NB], const int (&b)[NB], int co,
                                                    const float4 (&v)[NB], float* mom) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool has_bias = a.bias != nullptr;
  // unconditional load (from the filter when there is no bias; the value is then unused): a load inside a branch is
  // waited for at the end of that branch, one more round trip in front of the gate / residual reads
  const float4 bv = ld4(has_bias ? a.bias + co : reinterpret_cast<const float*>(a.weight));
  // Outputs that cannot fit the 256 MiB Infinity Cache (the 256^2 levels of the UNets at batch 4, every large map at batch 32)
  // leave with non-temporal stores: the consumer has to fetch them from HBM anyway, and the hint keeps them from evicting what
  // can stay (same box, plain -> hinted: C2 17.07 / 17.10 -> 17.01 / 17.03 ms per denoise step, C5 32.25 / 32.28 -> 32.20 / 32.17).
  // Smaller outputs -- the token GEMMs of the transformers, whose consumer follows at once -- keep the default policy (with
  // the hint on everything DiT-B/2 lost 0.1 - 0.3 %: profiles/r05_stream_nt_ab.txt).
  const bool stream_out = (int64_t)a.batch * a.hout * a.wout * a.cout_s >= ((int64_t)64 << 20);
  float4 g[NB], r[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    g[i] = z;
    r[i] = z;
    if (n[i] >= 0) {
      if constexpr (GATE) g[i] = ld4(a.gate + (int64_t)b[i] * a.gate_bstride + co);
      if constexpr (RES == 1) r[i] = ld4_io<IO>(a.res, (int64_t)n[i] * a.cout_s + co);
      if constexpr (RES == 2) r[i] = ld4_io<IO>(a.res, epilogue_res_index(a, n[i], b[i]) * a.cout_s + co);
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    float4 f = v[i];
    if (has_bias) {  // a select, not a branch: x + 0 would turn -0 into +0
      f.x += bv.x;
      f.y += bv.y;
      f.z += bv.z;
      f.w += bv.w;
    }
    if constexpr (ACT == 1) {
      f.x = az_silu(f.x);
      f.y = az_silu(f.y);
      f.z = az_silu(f.z);
      f.w = az_silu(f.w);
    }
    if constexpr (ACT == 6) {  // SiLU of the sum with the residual (RES = 1, no gate)
      f = make_float4(az_silu(f.x + r[i].x), az_silu(f.y + r[i].y), az_silu(f.z + r[i].z), az_silu(f.w + r[i].w));
      if (n[i] >= 0) st4_io<IO>(a.dst, (int64_t)n[i] * a.cout_s + co, f);
      continue;
    }
    if constexpr (ACT == 4) {  // SwiGLU over interleaved pairs: half the channels (see epilogue_apply_store)
      if (n[i] >= 0) st2_io<IO>(a.dst, (int64_t)n[i] * (a.cout_s / 2) + co / 2, make_float2(f.x * az_silu(f.y), f.z * az_silu(f.w)));
      continue;
    }
    if constexpr (GATE) {
      f.x *= g[i].x;
      f.y *= g[i].y;
      f.z *= g[i].z;
      f.w *= g[i].w;
    }
    if constexpr (RES != 0) {
      f.x += r[i].x;
      f.y += r[i].y;
      f.z += r[i].z;
      f.w += r[i].w;
    }
    if constexpr (IO == 0) {
      if (n[i] >= 0) {
        float* d = a.dst + (int64_t)n[i] * a.cout_s + co;
        if (stream_out) az_st_stream(d, f);
        else *reinterpret_cast<float4*>(d) = f;
      }
    } else {
      f = round_io<IO>(f);  // (the moments below are those of the stored values)
      if (n[i] >= 0) st4_io<IO>(a.dst, (int64_t)n[i] * a.cout_s + co, f);
    }
    if constexpr (MOM) {  // GroupNorm statistics of the output come from here (gn_quads; the host admits no skipped pixel)
      if (i == 0) mom[0] = f.x;
      const float d0 = f.x - mom[0], d1 = f.y - mom[0], d2 = f.z - mom[0], d3 = f.w - mom[0];
      mom[1] += (d0 + d1) + (d2 + d3);
      mom[2] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
}

// act 5: the tile is part of a fused q | k | v projection ('(3 H C)' channels); q and k are prepared for the attention kernel
// here, once: RMS norm over the head's channels, learned gains, RoPE (azula/nn/attention.py:92-95).  A thread holds the channel
// quad `co` of pixels (tokens) n[i]; the D / 4 quads of a (token, head) sit on D / 4 adjacent lanes (the exchange buffer is read
// back with a pixel's 32 quads on 32 consecutive lanes and tiles start at multiples of 128 channels), so the sum of squares is a
// butterfly over 8 / 16 / 32 lanes.  Every lane takes part in the shuffles (v lanes and skipped pixels compute values they drop).
template <int NB, int IO = 0>
__device__ __forceinline__ void epilogue_batch_qk(const AzConvArgs& a, const int (&n)[NB], const int (&b)[NB], int co,
                                                  const float4 (&v)[NB]) {
  const int D = a.qk_head_dim, HC = a.qk_heads * D;
  const int which = co / HC;  // 0: q, 1: k, 2: v
  const int cw = co - which * HC;
  const int head = cw / D, d = cw - head * D;
  const bool qk = which < 2;
  const bool has_bias = a.bias != nullptr;
  const float4 bv = ld4(has_bias ? a.bias + co : reinterpret_cast<const float*>(a.weight));
  const float* gp = which == 0 ? a.qk_q_weight : a.qk_k_weight;
  const bool has_gain = qk && gp != nullptr;
  const float4 gw = ld4(has_gain ? gp + d : reinterpret_cast<const float*>(a.weight));
  const bool rope = qk && a.qk_rope_cos != nullptr;
  float2 rc[NB], rs[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    rc[i] = make_float2(1.f, 1.f);
    rs[i] = make_float2(0.f, 0.f);
    if (rope && n[i] >= 0) {
      const int64_t ro = ((int64_t)(n[i] - b[i] * a.qk_tokens) * a.qk_heads + head) * (D / 2) + d / 2;
      rc[i] = *reinterpret_cast<const float2*>(a.qk_rope_cos + ro);
      rs[i] = *reinterpret_cast<const float2*>(a.qk_rope_sin + ro);
    }
  }
  const float inv_d = 1.f / (float)D;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    float4 f = v[i];
    if (has_bias) {
      f.x += bv.x;
      f.y += bv.y;
      f.z += bv.z;
      f.w += bv.w;
    }
    if (a.qk_rmsnorm) {  // (wave-uniform)
      float ss = (f.x * f.x + f.y * f.y) + (f.z * f.z + f.w * f.w);
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      if (D >= 64) ss += __shfl_xor(ss, 8, 64);
      if (D >= 128) ss += __shfl_xor(ss, 16, 64);
      const float r = qk ? rsqrtf(ss * inv_d + a.qk_eps) : 1.f;
      f.x *= r;
      f.y *= r;
      f.z *= r;
      f.w *= r;
    }
    if (has_gain) {
      f.x *= gw.x;
      f.y *= gw.y;
      f.z *= gw.z;
      f.w *= gw.w;
    }
    if (rope) {
      const float r0 = f.x, i0 = f.y, r1 = f.z, i1 = f.w;
      f.x = r0 * rc[i].x - i0 * rs[i].x;
      f.y = r0 * rs[i].x + i0 * rc[i].x;
      f.z = r1 * rc[i].y - i1 * rs[i].y;
      f.w = r1 * rs[i].y + i1 * rc[i].y;
    }
    if (n[i] >= 0) st4_io<IO>(a.dst, (int64_t)n[i] * a.cout_s + co, f);
  }
}

// A batch of NB outputs of one thread (same channel quad `co`, pixels n[i] of images b[i]; n[i] < 0: skip): all gate /
// residual reads are issued first, then the NB stores.
template <int NB, bool MOM = false, int IO = 0>
__device__ __forceinline__ void epilogue_store_batch(const AzConvArgs& a, const int (&n)[NB], const int (&b)[NB], int co,
                                                     const float4 (&v)[NB], int64_t ws_slab, float* mom = nullptr) {
  if (a.splitk > 1) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (n[i] >= 0) *reinterpret_cast<float4*>(a.workspace + (ws_slab + n[i]) * a.cout_s + co) = v[i];
    return;
  }
  if (a.act == 4) return epilogue_batch_nhwc<NB, false, 4, false, 0, IO>(a, n, b, co, v, mom);
  if (a.act == 5) return epilogue_batch_qk<NB, IO>(a, n, b, co, v);
  if (a.act == 6 && !a.res_up && !a.res_bcast) return epilogue_batch_nhwc<NB, false, 6, false, 1, IO>(a, n, b, co, v, mom);
  if (!a.dst_nchw && a.act <= 1) {
    const int rk = a.res == nullptr ? 0 : (a.res_up || a.res_bcast) ? 2 : 1;
    switch ((a.act * 2 + (a.gate != nullptr ? 1 : 0)) * 3 + rk) {
#define AZ_EPI_CASE(ACT, GATE, RES) \
  case (ACT * 2 + GATE) * 3 + RES:  \
    return epilogue_batch_nhwc<NB, MOM, ACT, GATE != 0, RES, IO>(a, n, b, co, v, mom);
      AZ_EPI_CASE(0, 0, 0) AZ_EPI_CASE(0, 0, 1) AZ_EPI_CASE(0, 0, 2) AZ_EPI_CASE(0, 1, 0) AZ_EPI_CASE(0, 1, 1) AZ_EPI_CASE(0, 1, 2)
      AZ_EPI_CASE(1, 0, 0) AZ_EPI_CASE(1, 0, 1) AZ_EPI_CASE(1, 0, 2) AZ_EPI_CASE(1, 1, 0) AZ_EPI_CASE(1, 1, 1) AZ_EPI_CASE(1, 1, 2)
#undef AZ_EPI_CASE
    }
  }
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 bv = a.bias ? ld4(a.bias + co) : z;
  float4 g[NB], r[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    g[i] = z;
    r[i] = z;
    if (n[i] >= 0) epilogue_fetch<IO>(a, n[i], b[i], co, g[i], r[i]);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (n[i] >= 0) {
      const float4 f = epilogue_apply_store<IO>(a, n[i], b[i], co, v[i], bv, g[i], r[i]);
      if constexpr (MOM) {
        if (i == 0) mom[0] = f.x;
        const float d0 = f.x - mom[0], d1 = f.y - mom[0], d2 = f.z - mom[0], d3 = f.w - mom[0];
        mom[1] += (d0 + d1) + (d2 + d3);
        mom[2] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
}

template <int IO = 0>
__device__ __forceinline__ void epilogue_store(const AzConvArgs& a, int n, int co, float4 v) {
  epilogue_store_b<IO>(a, n, n / (a.hout * a.wout), co, v);
}


}  // namespace

// (global scope: it crosses translation units through azi_winograd_x3_launch)
// Launch parameters of the fused Winograd kernels (conv_winograd_kernel: 8-channel stages; conv_winograd_x3_kernel: 16-channel steps).
struct WinoP {
  AzConvArgs a;
  int npix;
  int tiles_h, tiles_w, ntiles;
  int nkc0, nkc1, nk;  // 8-channel chunks per source, total
  int kps;             // chunks per split
  int cblocks;         // ceil(cout_s / WC)
  int tblocks;         // ceil(ntiles / WT)
  int gt, gc;          // workgroup order: rectangles of gt tile blocks x gc cout blocks, tile block fastest inside (1, cblocks: cout fastest)
  float out_scale;     // f16x2 form of the x3 kernel: 1 / w_scale; the accumulators are multiplied by out_scale / (the activation scale) behind the K loop
};
