// K4: multi-head self-attention softmax(q k^T * scale) v in fp32 on gfx950 MFMA, plus the
// token-layout helpers of the ViT path (patchify / unpatchify).
//
// Replaces F.scaled_dot_product_attention after per-head q/k RMSNorm (azula/nn/attention.py:89-104)
// and the einsum-softmax-einsum of the ADM attention blocks (plugins/adm/_src/unet.py:338-345,
// 371-379; softmax in fp32 there too).  Flash-style: the T x T score matrix never exists.
//
// Work split: one workgroup = one (batch, head) x 128 query rows; each of its 4 waves owns 32
// queries.  K/V tiles of 64 keys are staged through LDS (row stride D+4 floats: fragment
// ds_read_b128 are bank-conflict-free) and shared by the 4 waves.
//
// Both contractions use v_mfma_f32_32x32x2_f32 in the TRANSPOSED orientation
//     S^T = K  Q^T   (A = keys,  B = queries)      O^T = V^T P^T   (A = V^T, B = P^T)
// so that a lane owns one QUERY column: the online-softmax max/sum are in-lane reductions over
// its 16 score registers plus one cross-half shuffle, and the P^T registers feed the second
// MFMA directly as its B operand -- no LDS round trip, no cross-lane movement of P.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QT = 128;  // queries per workgroup
constexpr int KT = 64;   // keys per LDS tile

struct AttnP {
  AzAttnArgs a;
};

// key index inside a 32-key tile held by accumulator register r of a lane in half h
__device__ __forceinline__ int key_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// channels the q / k RMS norm averages over: the head size, or the REAL head size of a head that the host zero-padded to an
// instantiated one (AzAttnArgs.norm_dim; the same division, so an unpadded head rounds as before)
template <int D>
__device__ __forceinline__ float az_norm_dim(const AzAttnArgs& a) { return a.norm_dim > 0 ? (float)a.norm_dim : (float)D; }

template <int D>
__global__ __launch_bounds__(256) void attention_kernel(AzAttnArgs a) {
  constexpr int DP = (D + 31) / 32 * 32;  // padded head dim (whole 32-wide O^T tiles)
  constexpr int DT = DP / 32;          // number of 32-wide output tiles
  constexpr int LS = DP + 4;           // LDS row stride (floats)
  constexpr int KJ = D / 8;            // groups of 8 head-dim values in the QK^T contraction
  __shared__ __attribute__((aligned(16))) float smem[2 * KT * LS];
  float* Ks = smem;
  float* Vs = smem + KT * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h2 = lane >> 5;  // half-wave
  const int ql = lane & 31;
  const int bh = blockIdx.y;
  const int b = bh / a.heads;
  const int hd = bh - b * a.heads;
  const int T = a.tokens;
  const int q0 = blockIdx.x * QT + wave * 32;
  const int qi = q0 + ql;  // this lane's query

  const float* qp = a.q + (int64_t)b * a.q_bstride + (int64_t)hd * a.q_hstride;
  const float* kp = a.k + (int64_t)b * a.k_bstride + (int64_t)hd * a.k_hstride;
  const float* vp = a.v + (int64_t)b * a.v_bstride + (int64_t)hd * a.v_hstride;
  const uint8_t* mrow = (a.mask != nullptr && qi < T)
                            ? a.mask + (int64_t)b * a.mask_bstride + (int64_t)hd * a.mask_hstride + (int64_t)qi * T
                            : nullptr;

  // cooperative K/V tile loader: thread -> (row = tid / CH + pass * rows_per_pass, 16-B chunk)
  constexpr int CH = D / 4;               // chunks per row
  // CH a power of two: 256 / CH rows per pass, a row's chunks on CH consecutive lanes.  Otherwise (D = 80:
  // CH = 20) each wave takes 64 / CH whole rows so that a row never straddles two waves (4 lanes idle).
  constexpr bool POW2 = (CH & (CH - 1)) == 0;
  constexpr int RW = 64 / CH;
  constexpr int RPP = POW2 ? 256 / CH : 4 * RW;  // rows per pass
  const int lc = POW2 ? tid % CH : lane % CH;
  const int lr = POW2 ? tid / CH : wave * RW + lane / CH;
  const bool lactive = POW2 || lane < RW * CH;

  // register prefetch of the raw K / V rows of the NEXT tile: issued once the current tile has been staged, in
  // flight under its MFMAs (the tile loop used to expose one global round trip per 64 keys and workgroup: 181 -> 173 us
  // on 64 x 12 heads x 256 tokens, although 136 VGPRs now allow 3 instead of 4 waves per SIMD)
  constexpr int NPS = (KT + RPP - 1) / RPP;
  float4 pk[NPS], pv[NPS];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int row = ps * RPP + lr;
      const int key = k0 + row;
      pk[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      pv[ps] = pk[ps];
      if (row < KT && lactive && key < T) {
        pk[ps] = *reinterpret_cast<const float4*>(kp + (int64_t)key * a.k_tstride + lc * 4);
        pv[ps] = *reinterpret_cast<const float4*>(vp + (int64_t)key * a.v_tstride + lc * 4);
      }
    }
  };
  fetch(0);  // issued before the Q rows are read: both global round trips of the prologue overlap

  // ---- Q fragment: lane holds q[qi][8*jj + 4*h2 + s], pre-multiplied by scale (* rms factor)
  float qf[KJ][4];
  {
    float ss = 0.f;
#pragma unroll
    for (int jj = 0; jj < KJ; ++jj) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qi < T) v = *reinterpret_cast<const float4*>(qp + (int64_t)qi * a.q_tstride + 8 * jj + 4 * h2);
      qf[jj][0] = v.x;
      qf[jj][1] = v.y;
      qf[jj][2] = v.z;
      qf[jj][3] = v.w;
      ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    // scores are produced in log2 units (log2(e) folded into the query scale) so that the softmax numerator is ONE
    // v_exp_f32 (2^x, <= 1 ulp) per score instead of the ~20-instruction libm expf: on gfx950 vector instructions do
    // NOT overlap with fp32 MFMAs on a SIMD (measured, DESIGN.md section 4: the two times add), so every VALU
    // instruction removed from this loop is matrix-pipe time gained.  Parity cost measured on the golden ViT / ADM /
    // JiT vectors: below 1e-6 of the output scale (the tests' bounds are unchanged).
    float f = a.scale * 1.4426950408889634f;
    if (a.qk_rmsnorm) {
      ss += __shfl_xor(ss, 32, 64);
      f *= rsqrtf(ss / az_norm_dim<D>(a) + a.eps);
    }
#pragma unroll
    for (int jj = 0; jj < KJ; ++jj)
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[jj][s] *= f;
    if (a.q_weight != nullptr) {  // learned per-channel gain of the q RMS norm
#pragma unroll
      for (int jj = 0; jj < KJ; ++jj) {
        const float4 w = *reinterpret_cast<const float4*>(a.q_weight + 8 * jj + 4 * h2);
        qf[jj][0] *= w.x;
        qf[jj][1] *= w.y;
        qf[jj][2] *= w.z;
        qf[jj][3] *= w.w;
      }
    }
    if (a.rope_cos != nullptr && qi < T) {  // rotate adjacent (re, im) channel pairs by theta[token][head][pair]
      const int64_t rbase = (int64_t)qi * a.heads * (D / 2) + (int64_t)hd * (D / 2);
#pragma unroll
      for (int jj = 0; jj < KJ; ++jj) {
        const float2 c = *reinterpret_cast<const float2*>(a.rope_cos + rbase + 4 * jj + 2 * h2);
        const float2 sn = *reinterpret_cast<const float2*>(a.rope_sin + rbase + 4 * jj + 2 * h2);
        const float r0 = qf[jj][0], i0 = qf[jj][1], r1 = qf[jj][2], i1 = qf[jj][3];
        qf[jj][0] = r0 * c.x - i0 * sn.x;
        qf[jj][1] = r0 * sn.x + i0 * c.x;
        qf[jj][2] = r1 * c.y - i1 * sn.y;
        qf[jj][3] = r1 * sn.y + i1 * c.y;
      }
    }
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int rr = 0; rr < KT; rr += RPP) {
      const int row = rr + lr;
      if (row < KT && lactive) {
        const int key = k0 + row;
        float4 kv = pk[rr / RPP], vv = pv[rr / RPP];
        if (a.qk_rmsnorm) {  // per-key RMS norm: CH consecutive lanes hold one row
          float ss = (kv.x * kv.x + kv.y * kv.y) + (kv.z * kv.z + kv.w * kv.w);
          if constexpr (POW2) {
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) ss += __shfl_xor(ss, o, 64);
          } else {  // groups of 4 lanes by butterfly, then the CH / 4 group sums of this row
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const int base = (lane / CH) * CH;
            float tot = 0.f;
#pragma unroll
            for (int j = 0; j < CH / 4; ++j) tot += __shfl(ss, base + 4 * j, 64);
            ss = tot;
          }
          const float f = rsqrtf(ss / az_norm_dim<D>(a) + a.eps);
          kv.x *= f;
          kv.y *= f;
          kv.z *= f;
          kv.w *= f;
        }
        if (a.k_weight != nullptr) {
          const float4 w = *reinterpret_cast<const float4*>(a.k_weight + lc * 4);
          kv.x *= w.x;
          kv.y *= w.y;
          kv.z *= w.z;
          kv.w *= w.w;
        }
        if (a.rope_cos != nullptr && key < T) {
          const int64_t ro = (int64_t)key * a.heads * (D / 2) + (int64_t)hd * (D / 2) + 2 * lc;
          const float2 c = *reinterpret_cast<const float2*>(a.rope_cos + ro);
          const float2 sn = *reinterpret_cast<const float2*>(a.rope_sin + ro);
          const float r0 = kv.x, i0 = kv.y, r1 = kv.z, i1 = kv.w;
          kv.x = r0 * c.x - i0 * sn.x;
          kv.y = r0 * sn.x + i0 * c.x;
          kv.z = r1 * c.y - i1 * sn.y;
          kv.w = r1 * sn.y + i1 * c.y;
        }
        *reinterpret_cast<float4*>(Ks + row * LS + lc * 4) = kv;
        *reinterpret_cast<float4*>(Vs + row * LS + lc * 4) = vv;
        if (D < DP && lc == 0) {  // zero the padded V columns once per row
#pragma unroll
          for (int c = D; c < DP; c += 4) *reinterpret_cast<float4*>(Vs + row * LS + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    __syncthreads();
    if (k0 + KT < T) fetch(k0 + KT);

#pragma unroll
    for (int sub = 0; sub < KT / 32; ++sub) {
      if (k0 + sub * 32 >= T) break;  // wave-uniform
      // ---- S^T tile (32 keys x 32 queries): A = K rows, B = Q columns
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const float* kr = Ks + (sub * 32 + ql) * LS + 4 * h2;
#pragma unroll
      for (int jj = 0; jj < KJ; ++jj) {
        const float4 kf = *reinterpret_cast<const float4*>(kr + 8 * jj);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[jj][0], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[jj][1], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[jj][2], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[jj][3], sacc, 0, 0, 0);
      }
      // ---- online softmax over this lane's query column (ragged last tile only: mask the keys past T)
      if (k0 + sub * 32 + 32 > T) {  // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + sub * 32 + key_of(r, h2) >= T) sacc[r] = -INFINITY;
      }
      if (mrow != nullptr) {  // boolean attention mask: one byte per (query, key)
        unsigned char mb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + sub * 32 + key_of(r, h2);
          mb[r] = key < T ? mrow[key] : (unsigned char)1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = mb[r] == 0 ? -INFINITY : sacc[r];
      }
      float mt = sacc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      // -inf only while every key seen so far is masked for this query: subtract 0 then (all terms are 2^(-inf) = 0)
      const float m_sub = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_sub);  // 2^(-inf) = 0 on the first tile
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(sacc[r] - m_sub);  // masked keys: 2^(-inf) = 0
        sacc[r] = pv;
        ls += pv;
      }
      ls += __shfl_xor(ls, 32, 64);
      l_run = l_run * alpha + ls;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
      // ---- O^T += V^T P^T: A = V^T (lane: d = ql + 32 t, key = key_of(r, h2)), B = P^T regs
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vr = Vs + (sub * 32 + key_of(r, h2)) * LS + ql;
#pragma unroll
        for (int t = 0; t < DT; ++t)
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32 * t], sacc[r], oacc[t], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds O[qi][32 t + 8 g + 4 h2 + (0..3)] in oacc[t][4g .. 4g+3]
  if (qi < T) {
    const float inv = 1.f / l_run;
    float* op = a.out + (int64_t)b * a.o_bstride + (int64_t)hd * a.o_hstride + (int64_t)qi * a.o_tstride;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h2;
        if (d < D)
          *reinterpret_cast<float4*>(op + d) = make_float4(oacc[t][4 * g] * inv, oacc[t][4 * g + 1] * inv,
                                                            oacc[t][4 * g + 2] * inv, oacc[t][4 * g + 3] * inv);
      }
  }
}

// =================================================================================================
// Half-precision-operand attention for backbones cast to bf16 / f16 (the conv / GEMM counterpart is
// conv_igemm_half_kernel): q, k, v and the output stay fp32 tensors; the q/k RMS norms, gains, RoPE and the online
// softmax run in fp32; the two contractions use v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation.
//   S^T = K Q^T : A = K rows from LDS (bf16 [key][D+8]), B = the lane's query (D/16 fragments of 8 values, registers)
//   O^T += V^T P^T : B = the lane's own probabilities -- registers 8s .. 8s+7 of the S^T tile are 8 keys of one
//       query, i.e. exactly one B fragment if the MFMA's k index is mapped to keys as the C layout orders them;
//       A = V^T: V sits in LDS ROW-major [key][d] (one 8-byte store per staged chunk) and a fragment is two
//       ds_read_b64_tr_b16 -- the hardware's 4 x 4 transpose of 16-bit elements inside a 16-lane group delivers channel d of 4
//       keys per read, and the two groups of 4 keys are exactly those of registers 8 s .. 8 s + 7 (keys (r&3) + 8 (r>>2) + 4 h).
// With the matrix work 16x cheaper the exponentials bound the kernel: the half mode uses the hardware exp2.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af16x4 __attribute__((ext_vector_type(4)));
typedef float af32x8 __attribute__((ext_vector_type(8)));
typedef float af32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
struct HalfT {
  using x8 = abf16x8;
  using x4 = abf16x4;
  using x1 = __bf16;
};
template <>
struct HalfT<true> {
  using x8 = af16x8;
  using x4 = af16x4;
  using x1 = _Float16;
};

template <bool F16>
__device__ __forceinline__ f32x16 mfma_half(typename HalfT<F16>::x8 a, typename HalfT<F16>::x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 as3_s4;
template <int D, bool F16>
__global__ __launch_bounds__(256) void attention_half_kernel(AzAttnArgs a) {
  using H8 = typename HalfT<F16>::x8;
  using H4 = typename HalfT<F16>::x4;
  using H1 = typename HalfT<F16>::x1;
  constexpr int DP = (D + 31) / 32 * 32;
  constexpr int DT = DP / 32;
  constexpr int KS = D / 16;        // K-steps of the QK^T contraction
  constexpr int KLS = D + 8;        // K tile row stride (2-byte elements)
  // V tile ROW-major [key][VLS], consumed through ds_read_b64_tr_b16 (see attention_x3_kernel): row stride = +-16 dwords (mod 64)
  constexpr int VLS = DP == 32 ? 32 : DP == 128 ? 160 : 96;
  __shared__ __attribute__((aligned(16))) unsigned short hsm[KT * KLS + KT * VLS];
  H1* Ks = reinterpret_cast<H1*>(hsm);
  H1* Vt = Ks + KT * KLS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h2 = lane >> 5;
  const int ql = lane & 31;
  const int bh = blockIdx.y;
  const int b = bh / a.heads;
  const int hd = bh - b * a.heads;
  const int T = a.tokens;
  const int qi = blockIdx.x * QT + wave * 32 + ql;

  // q / k / v / out: fp32 tensors, or (io_dtype = 1: a module cast to half precision keeps its activations in HBM in its own type)
  // tensors of the kernel's 2-byte type -- same element strides, 8-byte accesses of 4 values
  constexpr int HIO = F16 ? 2 : 1;
  const bool hio = a.io_dtype != 0;
  auto ld = [&](const float* base, int64_t idx) __attribute__((always_inline)) { return hio ? ld4_io<HIO>(base, idx) : ld4_io<0>(base, idx); };
  const int64_t qo = (int64_t)b * a.q_bstride + (int64_t)hd * a.q_hstride;
  const int64_t ko = (int64_t)b * a.k_bstride + (int64_t)hd * a.k_hstride;
  const int64_t vo = (int64_t)b * a.v_bstride + (int64_t)hd * a.v_hstride;
  const uint8_t* mrow = (a.mask != nullptr && qi < T)
                            ? a.mask + (int64_t)b * a.mask_bstride + (int64_t)hd * a.mask_hstride + (int64_t)qi * T
                            : nullptr;

  // zero V once: channels d >= D (padding of the last 32-wide output tile) are never written again
  if (D < DP)
    for (int e = tid; e < KT * VLS / 2; e += 256) reinterpret_cast<unsigned*>(Vt)[e] = 0u;

  // ---- Q fragments: lane holds q[qi][16 ks + 8 h2 + (0..7)], scaled (and RMS-normalised, gained, rotated) in fp32
  H8 qf[KS];
  {
    float qv[KS][8];
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qi < T) v = ld(a.q, qo + (int64_t)qi * a.q_tstride + 16 * ks + 8 * h2 + 4 * hh);
        qv[ks][4 * hh + 0] = v.x;
        qv[ks][4 * hh + 1] = v.y;
        qv[ks][4 * hh + 2] = v.z;
        qv[ks][4 * hh + 3] = v.w;
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
    float f = a.scale;
    if (a.qk_rmsnorm) {
      ss += __shfl_xor(ss, 32, 64);
      f *= rsqrtf(ss / az_norm_dim<D>(a) + a.eps);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * h2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        qv[ks][j] *= f;
        if (a.q_weight != nullptr) qv[ks][j] *= a.q_weight[d0 + j];
      }
      if (a.rope_cos != nullptr && qi < T) {
        const int64_t rb = (int64_t)qi * a.heads * (D / 2) + (int64_t)hd * (D / 2) + d0 / 2;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const float c = a.rope_cos[rb + pr], sn = a.rope_sin[rb + pr];
          const float re = qv[ks][2 * pr], im = qv[ks][2 * pr + 1];
          qv[ks][2 * pr] = re * c - im * sn;
          qv[ks][2 * pr + 1] = re * sn + im * c;
        }
      }
      const af32x8 v8 = {qv[ks][0], qv[ks][1], qv[ks][2], qv[ks][3], qv[ks][4], qv[ks][5], qv[ks][6], qv[ks][7]};
      qf[ks] = __builtin_convertvector(v8, H8);
    }
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  constexpr int CH = D / 4;
  constexpr bool POW2 = (CH & (CH - 1)) == 0;
  constexpr int RW = 64 / CH;
  constexpr int RPP = POW2 ? 256 / CH : 4 * RW;
  const int lc = POW2 ? tid % CH : lane % CH;
  const int lr = POW2 ? tid / CH : wave * RW + lane / CH;
  const bool lactive = POW2 || lane < RW * CH;
  constexpr float LOG2E = 1.4426950408889634f;

  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();  // previous tile fully consumed (and, first time, the V^T zero fill is complete)
#pragma unroll
    for (int rr = 0; rr < KT; rr += RPP) {
      const int row = rr + lr;
      if (row < KT && lactive) {
        const int key = k0 + row;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (key < T) {
          kv = ld(a.k, ko + (int64_t)key * a.k_tstride + lc * 4);
          vv = ld(a.v, vo + (int64_t)key * a.v_tstride + lc * 4);
        }
        if (a.qk_rmsnorm) {
          float ss = (kv.x * kv.x + kv.y * kv.y) + (kv.z * kv.z + kv.w * kv.w);
          if constexpr (POW2) {
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) ss += __shfl_xor(ss, o, 64);
          } else {
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const int base = (lane / CH) * CH;
            float tot = 0.f;
#pragma unroll
            for (int j = 0; j < CH / 4; ++j) tot += __shfl(ss, base + 4 * j, 64);
            ss = tot;
          }
          const float f = rsqrtf(ss / az_norm_dim<D>(a) + a.eps);
          kv.x *= f;
          kv.y *= f;
          kv.z *= f;
          kv.w *= f;
        }
        if (a.k_weight != nullptr) {
          const float4 w = *reinterpret_cast<const float4*>(a.k_weight + lc * 4);
          kv.x *= w.x;
          kv.y *= w.y;
          kv.z *= w.z;
          kv.w *= w.w;
        }
        if (a.rope_cos != nullptr && key < T) {
          const int64_t ro = (int64_t)key * a.heads * (D / 2) + (int64_t)hd * (D / 2) + 2 * lc;
          const float2 c = *reinterpret_cast<const float2*>(a.rope_cos + ro);
          const float2 sn = *reinterpret_cast<const float2*>(a.rope_sin + ro);
          const float r0 = kv.x, i0 = kv.y, r1 = kv.z, i1 = kv.w;
          kv.x = r0 * c.x - i0 * sn.x;
          kv.y = r0 * sn.x + i0 * c.x;
          kv.z = r1 * c.y - i1 * sn.y;
          kv.w = r1 * sn.y + i1 * c.y;
        }
        const af32x4 k4 = {kv.x, kv.y, kv.z, kv.w};
        *reinterpret_cast<H4*>(Ks + row * KLS + lc * 4) = __builtin_convertvector(k4, H4);
        const af32x4 v4 = {vv.x, vv.y, vv.z, vv.w};
        *reinterpret_cast<H4*>(Vt + row * VLS + lc * 4) = __builtin_convertvector(v4, H4);
      }
    }
    __syncthreads();

#pragma unroll
    for (int sub = 0; sub < KT / 32; ++sub) {
      if (k0 + sub * 32 >= T) break;
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const H1* kr = Ks + (sub * 32 + ql) * KLS + 8 * h2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sacc = mfma_half<F16>(*reinterpret_cast<const H8*>(kr + 16 * ks), qf[ks], sacc);
      if (k0 + sub * 32 + 32 > T) {  // ragged last tile (wave-uniform): keys past T
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + sub * 32 + key_of(r, h2) >= T) sacc[r] = -INFINITY;
      }
      if (mrow != nullptr) {  // boolean attention mask: one byte per (query, key)
        unsigned char mb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + sub * 32 + key_of(r, h2);
          mb[r] = key < T ? mrow[key] : (unsigned char)1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = mb[r] == 0 ? -INFINITY : sacc[r];
      }
      float mt = sacc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float alpha = m_run == -INFINITY ? 0.f : exp2f((m_run - m_new) * LOG2E);
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = sacc[r] == -INFINITY ? 0.f : exp2f((sacc[r] - m_new) * LOG2E);
        sacc[r] = pv;
        ls += pv;
      }
      ls += __shfl_xor(ls, 32, 64);
      l_run = l_run * alpha + ls;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const af32x8 p8 = {sacc[8 * s2 + 0], sacc[8 * s2 + 1], sacc[8 * s2 + 2], sacc[8 * s2 + 3],
                           sacc[8 * s2 + 4], sacc[8 * s2 + 5], sacc[8 * s2 + 6], sacc[8 * s2 + 7]};
        const H8 pb = __builtin_convertvector(p8, H8);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          // A = V^T: k slots 0 .. 3 = keys 16 s2 + 4 h2 + (0 .. 3), slots 4 .. 7 = the same + 8 -- the order the S^T registers
          // 8 s2 .. 8 s2 + 7 hold them: two transpose reads of 4 keys x 16 channels per 16-lane group
          as3_s4* pa = (as3_s4*)(Vt + (sub * 32 + 16 * s2 + 4 * h2 + ((lane & 15) >> 2)) * VLS + 32 * t + (lane & 16) + 4 * (lane & 3));
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pa);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pa + 2 * VLS);
          const H8 va = __builtin_bit_cast(H8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          oacc[t] = mfma_half<F16>(va, pb, oacc[t]);
        }
      }
    }
  }

  if (qi < T) {
    const float inv = 1.f / l_run;
    const int64_t oo = (int64_t)b * a.o_bstride + (int64_t)hd * a.o_hstride + (int64_t)qi * a.o_tstride;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h2;
        if (d < D) {
          const float4 o = make_float4(oacc[t][4 * g] * inv, oacc[t][4 * g + 1] * inv, oacc[t][4 * g + 2] * inv, oacc[t][4 * g + 3] * inv);
          if (hio) st4_io<HIO>(a.out, oo + d, o);
          else st4_io<0>(a.out, oo + d, o);
        }
      }
  }
}

// =================================================================================================
// fp32 attention on the bf16 matrix pipe ("bf16x3", the default mode of fp32 modules -- engine.FP32_MFMA): q, k, v and the
// probabilities are split EXACTLY into three bf16 pieces each (az_split3) and every product of the two contractions is the six
// largest of the nine partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the dropped terms are at the level
// of one fp32 rounding of the product -- at 6 x 32 cycles per 16 values against 8 x 64 for v_mfma_f32_32x32x2_f32.
// Layout and fragment maps are attention_half_kernel's with three planes per operand: K [piece][key][D + 8], V [piece][key][row
// stride] read transposed by ds_read_b64_tr_b16, q pieces and probability pieces in registers; norms, gains, RoPE, the online softmax (scores in log2 units, hardware exp2) stay fp32 as in attention_kernel.
// H2: the "f16x2" form (az_attention_f16x2_f32; include/azula_amd.h: az_conv2d_f16x2_f32): THREE products per contraction on
// v_mfma_f32_32x32x16_f16.  S^T = K Q^T: the keys take the three-piece role [kh | kl | kh / 2^11] of k * 2^4 (az_split_w2h), the
// queries the two-piece role [h | l] of q / 2^4 (az_split2h) -- the scales cancel; O^T = V^T P^T: the probabilities (at most 2^8 under
// the lazy maximum) the three-piece role of p * 2^6, the values the two-piece role of v / 2^4 -- 1 / 4 folded into the final 1 / l.
// K keeps three LDS planes, V two; q two fragments per 16 channels.  Domain: |k| < 4094, |v|, |q * scale * log2 e| < 1.0e6 (beyond: NaN).
template <int D, int NW, bool H2 = false>  // NW = waves (32 queries each) per workgroup: 4, or 8 where the grid still fills the chip (the K / V tile is split and staged once per workgroup)
__global__ __launch_bounds__(64 * NW) void attention_x3_kernel(AzAttnArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int DP = (D + 31) / 32 * 32;
  constexpr int DT = DP / 32;
  constexpr int KS = D / 16;        // K-steps of the QK^T contraction
  constexpr int KLS = D + 8;        // K tile row stride (2-byte elements)
  // V tile: ROW-major [key][VLS] like K, consumed through ds_read_b64_tr_b16 (the hardware's 4 x 4 transpose of 16-bit elements
  // inside a 16-lane group: lane i supplies the address of 4 channels of key i / 4 and receives channel i of 4 keys).  Row stride
  // = +-16 dwords (mod 64): the 4 keys x 2 channel blocks that a 32-lane half reads in one access fall on eight distinct bank octets
  constexpr int VLS = DP == 32 ? 32 : DP == 128 ? 160 : 96;
  constexpr int KPL = KT * KLS, VPL = KT * VLS;  // elements per piece plane
  constexpr int NVP = H2 ? 2 : 3;  // V planes
  __shared__ __attribute__((aligned(16))) unsigned short xsm[3 * KPL + NVP * VPL];
  unsigned short* Ks = xsm;
  unsigned short* Vt = xsm + 3 * KPL;
  auto mma = [](const abf16x8& x, const abf16x8& y, const f32x16& c) __attribute__((always_inline)) {  // one partial product on the mode's pipe
    if constexpr (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(af16x8, x), __builtin_bit_cast(af16x8, y), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h2 = lane >> 5;
  const int ql = lane & 31;
  const int bh = blockIdx.y;
  const int b = bh / a.heads;
  const int hd = bh - b * a.heads;
  const int T = a.tokens;
  const int qi = blockIdx.x * (32 * NW) + wave * 32 + ql;

  const float* qp = a.q + (int64_t)b * a.q_bstride + (int64_t)hd * a.q_hstride;
  const float* kp = a.k + (int64_t)b * a.k_bstride + (int64_t)hd * a.k_hstride;
  const float* vp = a.v + (int64_t)b * a.v_bstride + (int64_t)hd * a.v_hstride;
  const uint8_t* mrow = (a.mask != nullptr && qi < T)
                            ? a.mask + (int64_t)b * a.mask_bstride + (int64_t)hd * a.mask_hstride + (int64_t)qi * T
                            : nullptr;

  // cooperative K/V tile loader (attention_kernel's): thread -> (row = tid / CH + pass * rows per pass, 16-byte chunk)
  constexpr int CH = D / 4;
  constexpr bool POW2 = (CH & (CH - 1)) == 0;
  constexpr int RW = 64 / CH;
  constexpr int RPP = POW2 ? NT / CH : NW * RW;
  const int lc = POW2 ? tid % CH : lane % CH;
  const int lr = POW2 ? tid / CH : wave * RW + lane / CH;
  const bool lactive = POW2 || lane < RW * CH;
  constexpr int NPS = (KT + RPP - 1) / RPP;
  float4 pk[NPS], pv[NPS];  // raw rows of the NEXT tile, in flight under the current tile's MFMAs
  auto fetch = [&](int k0) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int row = ps * RPP + lr;
      const int key = k0 + row;
      pk[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      pv[ps] = pk[ps];
      if (row < KT && lactive && key < T) {
        pk[ps] = *reinterpret_cast<const float4*>(kp + (int64_t)key * a.k_tstride + lc * 4);
        pv[ps] = *reinterpret_cast<const float4*>(vp + (int64_t)key * a.v_tstride + lc * 4);
      }
    }
  };
  fetch(0);

  // zero V once: channels d >= D (padding of the last 32-wide output tile) are never written again
  if (D < DP)
    for (int e = tid; e < NVP * VPL / 2; e += NT) reinterpret_cast<unsigned*>(Vt)[e] = 0u;

  // ---- Q fragments: lane holds q[qi][16 ks + 8 h2 + (0..7)], scaled to log2 units (and RMS-normalised, gained, rotated) in
  // fp32, then split: qf[piece][ks]
  abf16x8 qf[H2 ? 2 : 3][KS];
  {
    float qv[KS][8];
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qi < T) v = *reinterpret_cast<const float4*>(qp + (int64_t)qi * a.q_tstride + 16 * ks + 8 * h2 + 4 * hh);
        qv[ks][4 * hh + 0] = v.x;
        qv[ks][4 * hh + 1] = v.y;
        qv[ks][4 * hh + 2] = v.z;
        qv[ks][4 * hh + 3] = v.w;
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
    float f = a.scale * 1.4426950408889634f;  // (scores in log2 units: one v_exp_f32 per score, see attention_kernel)
    if (a.qk_rmsnorm) {
      ss += __shfl_xor(ss, 32, 64);
      f *= rsqrtf(ss / az_norm_dim<D>(a) + a.eps);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * h2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        qv[ks][j] *= f;
        if (a.q_weight != nullptr) qv[ks][j] *= a.q_weight[d0 + j];
      }
      if (a.rope_cos != nullptr && qi < T) {
        const int64_t rb = (int64_t)qi * a.heads * (D / 2) + (int64_t)hd * (D / 2) + d0 / 2;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const float c = a.rope_cos[rb + pr], sn = a.rope_sin[rb + pr];
          const float re = qv[ks][2 * pr], im = qv[ks][2 * pr + 1];
          qv[ks][2 * pr] = re * c - im * sn;
          qv[ks][2 * pr + 1] = re * sn + im * c;
        }
      }
      unsigned q3[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (H2) az_split2h(qv[ks][2 * j], qv[ks][2 * j + 1], q3[0][j], q3[1][j]);
        else az_split3(qv[ks][2 * j], qv[ks][2 * j + 1], q3[0][j], q3[1][j], q3[2][j]);
      }
#pragma unroll
      for (int pl = 0; pl < (H2 ? 2 : 3); ++pl) qf[pl][ks] = __builtin_bit_cast(abf16x8, make_uint4(q3[pl][0], q3[pl][1], q3[pl][2], q3[pl][3]));
    }
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  constexpr int PA[6] = {2, 1, 0, 1, 0, 0};  // (A piece, B piece) of the six partial products, smallest first
  constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
  // H2, three products: S^T (A = key planes [kh | kl | khs], B = query pieces [h | l]) and O^T (A = value planes [h | l], B = probability pieces [ph | pl | phs])
  constexpr int SA[3] = {1, 2, 0}, SB[3] = {0, 1, 0};
  constexpr int OA[3] = {0, 1, 0}, OB[3] = {1, 2, 0};
  constexpr float inv_rescale = H2 ? 0.25f : 1.f;  // O^T accumulates (v / 2^4) (p 2^6)

  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();  // previous tile fully consumed (and, first time, the V^T zero fill is complete)
#pragma unroll
    for (int rr = 0; rr < KT; rr += RPP) {
      const int row = rr + lr;
      if (row < KT && lactive) {
        const int key = k0 + row;
        float4 kv = pk[rr / RPP], vv = pv[rr / RPP];
        if (a.qk_rmsnorm) {
          float ss = (kv.x * kv.x + kv.y * kv.y) + (kv.z * kv.z + kv.w * kv.w);
          if constexpr (POW2) {
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) ss += __shfl_xor(ss, o, 64);
          } else {
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const int base = (lane / CH) * CH;
            float tot = 0.f;
#pragma unroll
            for (int j = 0; j < CH / 4; ++j) tot += __shfl(ss, base + 4 * j, 64);
            ss = tot;
          }
          const float f = rsqrtf(ss / az_norm_dim<D>(a) + a.eps);
          kv.x *= f;
          kv.y *= f;
          kv.z *= f;
          kv.w *= f;
        }
        if (a.k_weight != nullptr) {
          const float4 w = *reinterpret_cast<const float4*>(a.k_weight + lc * 4);
          kv.x *= w.x;
          kv.y *= w.y;
          kv.z *= w.z;
          kv.w *= w.w;
        }
        if (a.rope_cos != nullptr && key < T) {
          const int64_t ro = (int64_t)key * a.heads * (D / 2) + (int64_t)hd * (D / 2) + 2 * lc;
          const float2 c = *reinterpret_cast<const float2*>(a.rope_cos + ro);
          const float2 sn = *reinterpret_cast<const float2*>(a.rope_sin + ro);
          const float r0 = kv.x, i0 = kv.y, r1 = kv.z, i1 = kv.w;
          kv.x = r0 * c.x - i0 * sn.x;
          kv.y = r0 * sn.x + i0 * c.x;
          kv.z = r1 * c.y - i1 * sn.y;
          kv.w = r1 * sn.y + i1 * c.y;
        }
        unsigned k3[3][2], v3[3][2];
        if constexpr (H2) {
          az_split_w2h(kv.x, kv.y, 16.f, k3[0][0], k3[1][0], k3[2][0]);
          az_split_w2h(kv.z, kv.w, 16.f, k3[0][1], k3[1][1], k3[2][1]);
          az_split2h(vv.x, vv.y, v3[0][0], v3[1][0]);
          az_split2h(vv.z, vv.w, v3[0][1], v3[1][1]);
        } else {
          az_split3(kv.x, kv.y, k3[0][0], k3[1][0], k3[2][0]);
          az_split3(kv.z, kv.w, k3[0][1], k3[1][1], k3[2][1]);
          az_split3(vv.x, vv.y, v3[0][0], v3[1][0], v3[2][0]);
          az_split3(vv.z, vv.w, v3[0][1], v3[1][1], v3[2][1]);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          *reinterpret_cast<uint2*>(Ks + pl * KPL + row * KLS + lc * 4) = make_uint2(k3[pl][0], k3[pl][1]);
          if (pl < NVP) *reinterpret_cast<uint2*>(Vt + pl * VPL + row * VLS + lc * 4) = make_uint2(v3[pl][0], v3[pl][1]);
        }
      }
    }
    __syncthreads();
    if (k0 + KT < T) fetch(k0 + KT);

#pragma unroll
    for (int sub = 0; sub < KT / 32; ++sub) {
      if (k0 + sub * 32 >= T) break;
      // ---- S^T tile (32 keys x 32 queries): A = K pieces from LDS, B = the lane's q pieces
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const unsigned short* kr = Ks + (sub * 32 + ql) * KLS + 8 * h2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        abf16x8 kf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const abf16x8*>(kr + pl * KPL + 16 * ks);
#pragma unroll
        for (int t = 0; t < (H2 ? 3 : 6); ++t) {
          if constexpr (H2) sacc = mma(kf[SA[t]], qf[SB[t]][ks], sacc);
          else sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PA[t]], qf[PB[t]][ks], sacc, 0, 0, 0);
        }
      }
      if (k0 + sub * 32 + 32 > T) {  // ragged last tile (wave-uniform): keys past T
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + sub * 32 + key_of(r, h2) >= T) sacc[r] = -INFINITY;
      }
      if (mrow != nullptr) {  // boolean attention mask: one byte per (query, key)
        unsigned char mb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + sub * 32 + key_of(r, h2);
          mb[r] = key < T ? mrow[key] : (unsigned char)1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = mb[r] == 0 ? -INFINITY : sacc[r];
      }
      float mt = sacc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      // LAZY running maximum: the reference point m_run of a query moves only when the tile's maximum exceeds it by more than
      // 2^8 (scores are in log2 units), so the probabilities are at most 256 instead of at most 1 -- the same relative accuracy
      // in fp32, exact splits either way, the final O / l unchanged -- and the rescale of the 32 output accumulators (vector
      // instructions that do not overlap with the matrix pipe) runs in the first tile and after a genuine jump only:
      // wave-uniform branch, the lanes that stay put rescale by exactly 1
      const bool jump = mt > m_run + 8.f;  // (first tile: m_run = -inf; a fully masked tile: mt = -inf, no jump)
      if (__builtin_amdgcn_ballot_w64(jump) != 0ull) {
        const float m_new = jump ? mt : m_run;
        const float m_sub_new = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_sub_new);  // (-inf - finite -> 0; equal -> 1)
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
      }
      const float m_sub = m_run == -INFINITY ? 0.f : m_run;  // (-inf only while every key so far is masked: all terms 2^-inf = 0)
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pe = __builtin_amdgcn_exp2f(sacc[r] - m_sub);
        sacc[r] = pe;
        ls += pe;
      }
      ls += __shfl_xor(ls, 32, 64);
      l_run += ls;
      // ---- O^T += V^T P^T: B = the pieces of the lane's probabilities (registers 8 s2 .. 8 s2 + 7 = one fragment), A = V^T pieces
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        unsigned p3[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (H2) az_split_w2h(sacc[8 * s2 + 2 * j], sacc[8 * s2 + 2 * j + 1], 64.f, p3[0][j], p3[1][j], p3[2][j]);
          else az_split3(sacc[8 * s2 + 2 * j], sacc[8 * s2 + 2 * j + 1], p3[0][j], p3[1][j], p3[2][j]);
        }
        abf16x8 pb[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) pb[pl] = __builtin_bit_cast(abf16x8, make_uint4(p3[pl][0], p3[pl][1], p3[pl][2], p3[pl][3]));
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          // A = V^T: k slots 0 .. 3 = keys 16 s2 + 4 h2 + (0 .. 3), slots 4 .. 7 = the same + 8 (the order the S^T registers 8 s2 ..
          // 8 s2 + 7 hold them): two transpose reads of 4 keys x 16 channels per 16-lane group
          abf16x8 va[3];
          const unsigned short* vr = Vt + (sub * 32 + 16 * s2 + 4 * h2 + ((lane & 15) >> 2)) * VLS + 32 * t + (lane & 16) + 4 * (lane & 3);
#pragma unroll
          for (int pl = 0; pl < NVP; ++pl) {
            const as3_s4* pa = (const as3_s4*)(vr + pl * VPL);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<as3_s4*>(pa));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<as3_s4*>(pa) + 2 * VLS);  // (+ 8 keys: 8 VLS elements = 2 VLS vectors of 4)
            va[pl] = __builtin_bit_cast(abf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
#pragma unroll
          for (int u = 0; u < (H2 ? 3 : 6); ++u) {
            if constexpr (H2) oacc[t] = mma(va[OA[u]], pb[OB[u]], oacc[t]);
            else oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[PA[u]], pb[PB[u]], oacc[t], 0, 0, 0);
          }
        }
      }
    }
  }

  if (qi < T) {
    const float inv = inv_rescale / l_run;
    float* op = a.out + (int64_t)b * a.o_bstride + (int64_t)hd * a.o_hstride + (int64_t)qi * a.o_tstride;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h2;
        if (d < D)
          *reinterpret_cast<float4*>(op + d) = make_float4(oacc[t][4 * g] * inv, oacc[t][4 * g + 1] * inv,
                                                            oacc[t][4 * g + 2] * inv, oacc[t][4 * g + 3] * inv);
      }
  }
}

// NCHW (B, Z, H, W) -> tokens (B, L = H/p * W/p, cs) with feature index z*p*p + a*p + b
// ('... Z (A a) (B b) -> ... A B (Z a b)', azula/nn/layers.py:198-222); optional scale.
// IDX = unsigned when the element count fits 32 bits (always on the sampling path): the index decomposition is a chain
// of divisions per element, and 64-bit ones cost ~40 instructions each (0.8 TB/s measured with int64_t throughout).
template <typename IDX>
__global__ __launch_bounds__(256) void patchify_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                       const float* __restrict__ scale, int64_t B, int Z, int H,
                                                       int W, int p, int cs) {
  const float s = scale ? *scale : 1.f;
  const IDX Hp = H / p, Wp = W / p, F = Z * p * p, pp = p * p;
  const IDX total = (IDX)(B * Hp * Wp * cs);
  for (IDX e = (IDX)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (IDX)gridDim.x * blockDim.x) {
    const IDX f = e % (IDX)cs;
    const IDX tok = e / (IDX)cs;
    float v = 0.f;
    if (f < F) {
      const IDX z = f / pp, ab = f - z * pp, ai = ab / (IDX)p, bi = ab - ai * p;
      const IDX wp = tok % Wp, hp = (tok / Wp) % Hp;
      const IDX b = tok / (Wp * Hp);
      v = az_mul(s, src[((int64_t)(b * Z + z) * H + hp * p + ai) * W + wp * p + bi]);
    }
    dst[e] = v;
  }
}
template <typename IDX>
__global__ __launch_bounds__(256) void unpatchify_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                         int64_t B, int Z, int H, int W, int p, int cs) {
  const IDX Hp = H / p, Wp = W / p;
  const IDX total = (IDX)(B * Z * H * W);
  for (IDX e = (IDX)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (IDX)gridDim.x * blockDim.x) {
    const IDX w = e % (IDX)W, r1 = e / (IDX)W;
    const IDX hh = r1 % (IDX)H, r2 = r1 / (IDX)H;
    const IDX z = r2 % (IDX)Z, b = r2 / (IDX)Z;
    const IDX wp = w / (IDX)p, bi = w - wp * p, hp = hh / (IDX)p, ai = hh - hp * p;
    const int64_t tok = (int64_t)(b * Hp + hp) * Wp + wp;
    dst[e] = src[tok * cs + z * p * p + ai * p + bi];
  }
}

// Token-window copy: dst[b, dst_off + i, :] = src[b, src_off + i, :], i < n (16-byte vectors).
__global__ __launch_bounds__(256) void token_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src,
                                                         int64_t dst_tokens, int64_t dst_off, int64_t src_tokens,
                                                         int64_t src_off, int64_t n, int64_t B, int64_t cs4) {
  const int64_t total = B * n * cs4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = e % cs4, i = (e / cs4) % n, b = e / (cs4 * n);
    dst[((b * dst_tokens + dst_off + i) * cs4) + c] = src[((b * src_tokens + src_off + i) * cs4) + c];
  }
}

// dst[b, dst_off + j, :] = row[b, :] + pos[j, :], j < n  (in-context class tokens of JiT,
// plugins/jit/_src/model.py:364-367: y_emb repeated over the context plus its positional embedding)
__global__ __launch_bounds__(256) void token_fill_kernel(float4* __restrict__ dst, const float4* __restrict__ row,
                                                         const float4* __restrict__ pos, int64_t dst_tokens,
                                                         int64_t dst_off, int64_t n, int64_t B, int64_t cs4,
                                                         int64_t row_bstride4) {
  const int64_t total = B * n * cs4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = e % cs4, j = (e / cs4) % n, b = e / (cs4 * n);
    const float4 r = row[b * row_bstride4 + c], q = pos[j * cs4 + c];
    dst[((b * dst_tokens + dst_off + j) * cs4) + c] =
        make_float4(az_add(r.x, q.x), az_add(r.y, q.y), az_add(r.z, q.z), az_add(r.w, q.w));
  }
}

// The same with the destination in a 2-byte type (IO 1: bfloat16, 2: IEEE half; row / pos stay fp32): the in-context class tokens of
// a JiT cast to half precision.
template <int IO>
__global__ __launch_bounds__(256) void token_fill_h16_kernel(float* __restrict__ dst, const float4* __restrict__ row,
                                                             const float4* __restrict__ pos, int64_t dst_tokens, int64_t dst_off,
                                                             int64_t n, int64_t B, int64_t cs4, int64_t row_bstride4) {
  const int64_t total = B * n * cs4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = e % cs4, j = (e / cs4) % n, b = e / (cs4 * n);
    const float4 r = row[b * row_bstride4 + c], q = pos[j * cs4 + c];
    st4_io<IO>(dst, (((b * dst_tokens + dst_off + j) * cs4) + c) * 4, make_float4(az_add(r.x, q.x), az_add(r.y, q.y), az_add(r.z, q.z), az_add(r.w, q.w)));
  }
}

// Sinusoidal timestep embedding, cos block then sin block (plugins/jit/_src/model.py:59-81):
// dst[r, j] = cos(t_r f_j), dst[r, half + j] = sin(t_r f_j), f_j = exp(-ln(max_period) j / half).
__global__ __launch_bounds__(256) void timestep_embedding_kernel(float* __restrict__ dst, int64_t ldd,
                                                                 const float* __restrict__ t, int64_t t_stride,
                                                                 int64_t rows, int half, float log_period) {
  const int64_t total = rows * half;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(e % half);
    const int64_t r = e / half;
    const float f = expf(az_mul(-log_period, (float)j) / (float)half);
    const float arg = az_mul(t[r * t_stride], f);
    dst[r * ldd + j] = cosf(arg);
    dst[r * ldd + half + j] = sinf(arg);
  }
}

// y[r, c] = x[r, 2c] * silu(x[r, 2c + 1])  (azula/nn/layers.py:107-110: unflatten(-1, (-1, 2)))
__global__ __launch_bounds__(256) void swiglu_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t rows,
                                                     int cout, int xs, int ys) {
  const int64_t total = rows * ys;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % ys);
    const int64_t r = e / ys;
    float v = 0.f;
    if (c < cout) {
      const float2 p = *reinterpret_cast<const float2*>(x + r * xs + 2 * c);
      v = p.x * az_silu(p.y);
    }
    y[e] = v;
  }
}

}  // namespace

template <bool F16>
static int attention_half_launch(const AzAttnArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->q && a->k && a->v && a->out, AZ_E_NULL);
  AZ_REQUIRE(a->io_dtype == 0 || a->io_dtype == 1, AZ_E_SHAPE);
  AZ_REQUIRE(a->batch > 0 && a->heads > 0 && a->tokens > 0, AZ_E_SHAPE);
  AZ_REQUIRE(a->head_dim == 16 || a->head_dim == 32 || a->head_dim == 64 || a->head_dim == 80 || a->head_dim == 128,
             AZ_E_UNSUPPORTED);
  AZ_REQUIRE(a->norm_dim >= 0 && a->norm_dim <= a->head_dim, AZ_E_SHAPE);
  if (a->io_dtype) AZ_REQUIRE((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) & 7u) == 0, AZ_E_ALIGN);
  else AZ_REQUIRE(AZ_ALIGNED16(a->q) && AZ_ALIGNED16(a->k) && AZ_ALIGNED16(a->v) && AZ_ALIGNED16(a->out), AZ_E_ALIGN);
  const int64_t strides[] = {a->q_bstride, a->q_tstride, a->q_hstride, a->k_bstride, a->k_tstride, a->k_hstride,
                             a->v_bstride, a->v_tstride, a->v_hstride, a->o_bstride, a->o_tstride, a->o_hstride};
  for (int64_t s : strides) AZ_REQUIRE(s % 4 == 0, AZ_E_ALIGN);
  dim3 grid((unsigned)((a->tokens + QT - 1) / QT), (unsigned)(a->batch * a->heads));
  hipStream_t st = az_s(stream);
  switch (a->head_dim) {
    case 16: hipLaunchKernelGGL((attention_half_kernel<16, F16>), grid, dim3(256), 0, st, *a); break;
    case 32: hipLaunchKernelGGL((attention_half_kernel<32, F16>), grid, dim3(256), 0, st, *a); break;
    case 64: hipLaunchKernelGGL((attention_half_kernel<64, F16>), grid, dim3(256), 0, st, *a); break;
    case 80: hipLaunchKernelGGL((attention_half_kernel<80, F16>), grid, dim3(256), 0, st, *a); break;
    default: hipLaunchKernelGGL((attention_half_kernel<128, F16>), grid, dim3(256), 0, st, *a); break;
  }
  return az_launch_status();
}

extern "C" {

int az_attention_bf16_f32(const AzAttnArgs* a, az_stream_t stream) { return attention_half_launch<false>(a, stream); }
int az_attention_f16_f32(const AzAttnArgs* a, az_stream_t stream) { return attention_half_launch<true>(a, stream); }

/* az_attention_f32 with both contractions evaluated as 3 x bf16 operand pieces / 6 partial products on the bf16 MFMA (fp32
 * accumulation, fp32 softmax): fp32-level accuracy at 0.375 x the matrix-pipe time.  head_dim 16, 32, 64, 80 or 128.       */
static int attention_x3_launch(const AzAttnArgs* a, az_stream_t stream, bool h2);
int az_attention_x3_f32(const AzAttnArgs* a, az_stream_t stream) { return attention_x3_launch(a, stream, false); }
/* The f16x2 form of the same kernel (include/azula_amd.h): q / v as two IEEE half pieces, k / the probabilities as three, three
 * partial products per contraction on v_mfma_f32_32x32x16_f16.  Domain: |k| < 4094, |v| and |q * scale * log2 e| < 1.0e6.        */
int az_attention_f16x2_f32(const AzAttnArgs* a, az_stream_t stream) { return attention_x3_launch(a, stream, true); }

static int attention_x3_launch(const AzAttnArgs* a, az_stream_t stream, bool h2) {
  AZ_REQUIRE(a && a->q && a->k && a->v && a->out, AZ_E_NULL);
  AZ_REQUIRE(a->io_dtype == 0, AZ_E_UNSUPPORTED);  // (half-precision tensors: the bf16 / f16 entries)
  AZ_REQUIRE(a->batch > 0 && a->heads > 0 && a->tokens > 0, AZ_E_SHAPE);
  AZ_REQUIRE(a->head_dim == 16 || a->head_dim == 32 || a->head_dim == 64 || a->head_dim == 80 || a->head_dim == 128,
             AZ_E_UNSUPPORTED);
  AZ_REQUIRE(a->norm_dim >= 0 && a->norm_dim <= a->head_dim, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(a->q) && AZ_ALIGNED16(a->k) && AZ_ALIGNED16(a->v) && AZ_ALIGNED16(a->out), AZ_E_ALIGN);
  const int64_t strides[] = {a->q_bstride, a->q_tstride, a->q_hstride, a->k_bstride, a->k_tstride, a->k_hstride,
                             a->v_bstride, a->v_tstride, a->v_hstride, a->o_bstride, a->o_tstride, a->o_hstride};
  for (int64_t s : strides) AZ_REQUIRE(s % 4 == 0, AZ_E_ALIGN);
  hipStream_t st = az_s(stream);
  // 256 queries per workgroup (8 waves) halve the splitting / staging of K and V per query; taken where whole 256-query blocks
  // still make two rounds of the 256 CUs (one 8-wave workgroup is resident per CU).  Measured (us, 4 / 8 waves): 64 x 12 heads x
  // 256 tokens 111 / 104; smaller grids lose: 4 x 8 x 1024 82 / 90, 4 x 16 x 256 22 / 26; 288 tokens (not whole blocks) 95 / 104
  const int64_t bh = (int64_t)a->batch * a->heads;
  const bool wide = a->head_dim <= 80 && a->tokens % 256 == 0 && bh * (a->tokens / 256) >= 512;
  const int qt = wide ? 256 : QT;
  dim3 grid((unsigned)((a->tokens + qt - 1) / qt), (unsigned)bh);
#define AZ_ATT_X3(DD)                                                                                         \
  if (wide && h2) hipLaunchKernelGGL((attention_x3_kernel<DD, 8, true>), grid, dim3(512), 0, st, *a);         \
  else if (wide) hipLaunchKernelGGL((attention_x3_kernel<DD, 8>), grid, dim3(512), 0, st, *a);                \
  else if (h2) hipLaunchKernelGGL((attention_x3_kernel<DD, 4, true>), grid, dim3(256), 0, st, *a);            \
  else hipLaunchKernelGGL((attention_x3_kernel<DD, 4>), grid, dim3(256), 0, st, *a)
  switch (a->head_dim) {
    case 16: AZ_ATT_X3(16); break;
    case 32: AZ_ATT_X3(32); break;
    case 64: AZ_ATT_X3(64); break;
    case 80: AZ_ATT_X3(80); break;
    default:
      if (h2) hipLaunchKernelGGL((attention_x3_kernel<128, 4, true>), grid, dim3(256), 0, st, *a);
      else hipLaunchKernelGGL((attention_x3_kernel<128, 4>), grid, dim3(256), 0, st, *a);
      break;
  }
#undef AZ_ATT_X3
  return az_launch_status();
}

int az_attention_f32(const AzAttnArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->q && a->k && a->v && a->out, AZ_E_NULL);
  AZ_REQUIRE(a->io_dtype == 0, AZ_E_UNSUPPORTED);  // (half-precision tensors: the bf16 / f16 entries)
  AZ_REQUIRE(a->batch > 0 && a->heads > 0 && a->tokens > 0, AZ_E_SHAPE);
  AZ_REQUIRE(a->head_dim == 8 || a->head_dim == 16 || a->head_dim == 32 || a->head_dim == 64 || a->head_dim == 80 ||
                 a->head_dim == 128,
             AZ_E_UNSUPPORTED);
  AZ_REQUIRE(a->norm_dim >= 0 && a->norm_dim <= a->head_dim, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(a->q) && AZ_ALIGNED16(a->k) && AZ_ALIGNED16(a->v) && AZ_ALIGNED16(a->out), AZ_E_ALIGN);
  const int64_t strides[] = {a->q_bstride, a->q_tstride, a->q_hstride, a->k_bstride, a->k_tstride, a->k_hstride,
                             a->v_bstride, a->v_tstride, a->v_hstride, a->o_bstride, a->o_tstride, a->o_hstride};
  for (int64_t s : strides) AZ_REQUIRE(s % 4 == 0, AZ_E_ALIGN);
  dim3 grid((unsigned)((a->tokens + QT - 1) / QT), (unsigned)(a->batch * a->heads));
  hipStream_t st = az_s(stream);
  switch (a->head_dim) {
    case 8: hipLaunchKernelGGL(attention_kernel<8>, grid, dim3(256), 0, st, *a); break;
    case 16: hipLaunchKernelGGL(attention_kernel<16>, grid, dim3(256), 0, st, *a); break;
    case 32: hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(256), 0, st, *a); break;
    case 64: hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), 0, st, *a); break;
    case 80: hipLaunchKernelGGL(attention_kernel<80>, grid, dim3(256), 0, st, *a); break;
    default: hipLaunchKernelGGL(attention_kernel<128>, grid, dim3(256), 0, st, *a); break;
  }
  return az_launch_status();
}

int az_swiglu_f32(float* y, const float* x, int64_t rows, int64_t cout, int64_t xs, int64_t ys, az_stream_t stream) {
  AZ_REQUIRE(y && x, AZ_E_NULL);
  AZ_REQUIRE(rows > 0 && cout > 0 && xs >= 2 * cout && ys >= cout && xs % 2 == 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(swiglu_kernel, dim3(az_stream_grid(rows * ys, 256)), dim3(256), 0, az_s(stream), y, x, rows,
                     (int)cout, (int)xs, (int)ys);
  return az_launch_status();
}

int az_patchify_f32(float* dst, const float* src, const float* scale_dev, int64_t B, int64_t Z, int64_t H, int64_t W,
                    int64_t p, int64_t cs, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && Z > 0 && p > 0 && H % p == 0 && W % p == 0 && cs >= Z * p * p, AZ_E_SHAPE);
  const int64_t total = B * (H / p) * (W / p) * cs;
  // 32-bit index arithmetic whenever the sizes allow (the grid-stride increment must not wrap either)
  if (total < (1ll << 31) && B * Z * H * W < (1ll << 31))
    hipLaunchKernelGGL(patchify_kernel<unsigned>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream), dst, src,
                       scale_dev, B, (int)Z, (int)H, (int)W, (int)p, (int)cs);
  else
    hipLaunchKernelGGL(patchify_kernel<int64_t>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream), dst, src,
                       scale_dev, B, (int)Z, (int)H, (int)W, (int)p, (int)cs);
  return az_launch_status();
}

int az_unpatchify_f32(float* dst, const float* src, int64_t B, int64_t Z, int64_t H, int64_t W, int64_t p, int64_t cs,
                      az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && Z > 0 && p > 0 && H % p == 0 && W % p == 0 && cs >= Z * p * p, AZ_E_SHAPE);
  if (B * Z * H * W < (1ll << 31) && B * (H / p) * (W / p) * cs < (1ll << 31))
    hipLaunchKernelGGL(unpatchify_kernel<unsigned>, dim3(az_stream_grid(B * Z * H * W, 256)), dim3(256), 0, az_s(stream),
                       dst, src, B, (int)Z, (int)H, (int)W, (int)p, (int)cs);
  else
    hipLaunchKernelGGL(unpatchify_kernel<int64_t>, dim3(az_stream_grid(B * Z * H * W, 256)), dim3(256), 0, az_s(stream),
                       dst, src, B, (int)Z, (int)H, (int)W, (int)p, (int)cs);
  return az_launch_status();
}

int az_token_copy_f32(float* dst, int64_t dst_tokens, int64_t dst_off, const float* src, int64_t src_tokens,
                      int64_t src_off, int64_t n, int64_t B, int64_t cs, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && n > 0 && cs > 0 && cs % 4 == 0 && dst_off >= 0 && src_off >= 0 && dst_off + n <= dst_tokens &&
                 src_off + n <= src_tokens,
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(dst) && AZ_ALIGNED16(src), AZ_E_ALIGN);
  hipLaunchKernelGGL(token_copy_kernel, dim3(az_stream_grid(B * n * (cs / 4), 256)), dim3(256), 0, az_s(stream),
                     reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src), dst_tokens, dst_off,
                     src_tokens, src_off, n, B, cs / 4);
  return az_launch_status();
}

int az_token_fill_f32(float* dst, int64_t dst_tokens, int64_t dst_off, int64_t n, const float* row,
                      int64_t row_bstride, const float* pos, int64_t B, int64_t cs, az_stream_t stream) {
  AZ_REQUIRE(dst && row && pos, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && n > 0 && cs > 0 && cs % 4 == 0 && row_bstride % 4 == 0 && dst_off >= 0 &&
                 dst_off + n <= dst_tokens,
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(dst) && AZ_ALIGNED16(row) && AZ_ALIGNED16(pos), AZ_E_ALIGN);
  hipLaunchKernelGGL(token_fill_kernel, dim3(az_stream_grid(B * n * (cs / 4), 256)), dim3(256), 0, az_s(stream),
                     reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(row),
                     reinterpret_cast<const float4*>(pos), dst_tokens, dst_off, n, B, cs / 4, row_bstride / 4);
  return az_launch_status();
}

int az_token_fill_h16(void* dst, int64_t dst_tokens, int64_t dst_off, int64_t n, const float* row, int64_t row_bstride,
                      const float* pos, int64_t B, int64_t cs, int32_t dtype, az_stream_t stream) {
  AZ_REQUIRE(dst && row && pos, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && n > 0 && cs > 0 && cs % 8 == 0 && row_bstride % 4 == 0 && dst_off >= 0 && dst_off + n <= dst_tokens &&
                 (dtype == 1 || dtype == 2),
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(dst) && AZ_ALIGNED16(row) && AZ_ALIGNED16(pos), AZ_E_ALIGN);
  const dim3 grid(az_stream_grid(B * n * (cs / 4), 256));
  if (dtype == 1)
    hipLaunchKernelGGL(token_fill_h16_kernel<1>, grid, dim3(256), 0, az_s(stream), reinterpret_cast<float*>(dst), reinterpret_cast<const float4*>(row),
                       reinterpret_cast<const float4*>(pos), dst_tokens, dst_off, n, B, cs / 4, row_bstride / 4);
  else
    hipLaunchKernelGGL(token_fill_h16_kernel<2>, grid, dim3(256), 0, az_s(stream), reinterpret_cast<float*>(dst), reinterpret_cast<const float4*>(row),
                       reinterpret_cast<const float4*>(pos), dst_tokens, dst_off, n, B, cs / 4, row_bstride / 4);
  return az_launch_status();
}

int az_timestep_embedding_f32(float* dst, int64_t ldd, const float* t_dev, int64_t t_stride, int64_t rows,
                              int32_t half, float max_period, az_stream_t stream) {
  AZ_REQUIRE(dst && t_dev, AZ_E_NULL);
  AZ_REQUIRE(rows > 0 && half > 0 && ldd >= 2 * half && max_period > 0.f && (t_stride == 0 || t_stride == 1),
             AZ_E_SHAPE);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(az_stream_grid(rows * half, 256)), dim3(256), 0, az_s(stream),
                     dst, ldd, t_dev, t_stride, rows, (int)half, logf(max_period));
  return az_launch_status();
}

}  // extern "C"
