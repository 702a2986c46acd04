// Shared device/host helpers for the gfx950 kernels of azula_amd.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azula_amd.h"

#define AZ_WAVE 64

#define AZ_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

#define AZ_ALIGNED16(p) ((((uintptr_t)(p)) & 15u) == 0)

static inline int az_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AZ_OK : (int)e;
}

static inline hipStream_t az_s(az_stream_t s) { return (hipStream_t)s; }

// A/B switches of the launchers (kernel-selection overrides used by tools/ and a few tests) are honoured ONLY under the explicit
// debug switch AZ_DEBUG_AB: a stray variable in a production environment never steers kernel selection.
#include <stdlib.h>
static inline const char* az_ab_env(const char* name) { return getenv("AZ_DEBUG_AB") ? getenv(name) : nullptr; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set once per (kernel, device), lock-free and
// safe from several host threads (the launchers keep no other mutable state).  `mask` = one static std::atomic per kernel, bit d =
// device d done (devices >= 64 simply set the attribute on every launch: it is idempotent).
#include <atomic>
static inline hipError_t az_max_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& mask) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = (unsigned)dev < 64u ? (1ull << dev) : 0ull;
  if (bit && (mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && bit) mask.fetch_or(bit, std::memory_order_release);
  return e;
}

// Separately rounded fp32 ops: the compiler may not contract these into FMAs, so a chain
// written with them reproduces torch's eager op-by-op rounding bit for bit.
__device__ __forceinline__ float az_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float az_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float az_sub(float a, float b) { return __fsub_rn(a, b); }

// SiLU = v * sigmoid(v) as 5 vector instructions (every SiLU of the library: conv epilogues, affine_act, small linears,
// modulation MLPs).  v_exp_f32 and v_rcp_f32 are each within 1 ulp, but the fp32 rounding of v * log2(e) in front of the
// exponential gives exp(-v) a RELATIVE error of about |v| * 6e-8, so the bound is absolute, not relative: for v >= 0 the result
// is within ~3 ulp of v / (1 + expf(-v)); for negative v (result ~ v * exp(v), tiny) the relative error grows like |v| * 1e-7
// (about 17 ulp at v = -20) while the absolute error stays below 1e-7 * |v| * exp(v) <= 4e-8, and results whose reciprocal is
// subnormal flush to -0.  Limits: -inf -> NaN, very negative -> -0, +inf -> +inf.  The ocml expf + IEEE division form took
// ~28 instructions; in a conv epilogue (one workgroup per CU, nothing to hide behind) that was 8 k of a workgroup's 170 k
// cycles (profiles/r03_wino_timeline.txt), and vector instructions add to matrix time on this chip.
__device__ __forceinline__ float az_silu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

// 64-lane butterfly reductions (wave64).
__device__ __forceinline__ float az_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float az_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// x = x1 + x2 + x3 EXACTLY, each piece a bf16 (three 8-bit slices of the 24-bit significand, by truncation): the operand form of
// the "bf16x3" kernels, which evaluate a product as the six largest of the nine partial products on v_mfma_f32_32x32x16_bf16
// with fp32 accumulation (conv.hip, wino_x3.hip, attention.hip).  Two values per call; p1 / p2 / p3 = the packed (x1, x0) pairs of the pieces.
// Domain: FINITE inputs.  x = +-Inf gives Inf - Inf = NaN remainders (an Inf operand turns into NaN where the fp32 MFMA would
// propagate Inf; NaN stays NaN), and the low pieces of operands below ~2^-110 fall under the bf16 subnormal range: such
// products keep ~16 instead of 24 significant bits (their absolute error is below 2^-126, far under any accumulated sum of O(1)
// activations).  (The gfx950 matrix pipe HONOURS subnormal bf16 / fp16 operands -- tools/mfma_denorm_probe.hip, round 6; an
// earlier version of this comment assumed a flush.)  tests/test_gpu_kernels.py::test_x3_split_domain pins both statements.
__device__ __forceinline__ void az_split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
  const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u);  // exact: the low 16 significand bits
  const float r1 = x1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
  const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
  const float s0 = r0 - __builtin_bit_cast(float, v0 & 0xFFFF0000u);
  const float s1 = r1 - __builtin_bit_cast(float, v1 & 0xFFFF0000u);
  p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // high halves of (x1, x0) = truncated bf16 pair
  p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  p3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}

// The activation operand of the "f16x2" kernels (include/azula_amd.h: az_conv2d_f16x2_f32): x' = x * AZ_F16X2_IN_SCALE as two IEEE
// half pieces, h = fp16(x') (round to nearest: v_cvt_pk_f16_f32) and l = fp16((x' - h) * 2^11) -- x' = h + l / 2^11 to 22 - 23
// significant bits; the residual is exact in fp32, scaled by 2^11 so that it is a NORMAL half value wherever h is (the matrix pipe
// honours subnormal halves too: tools/mfma_denorm_probe.hip, so below 2^-14 the pieces degrade gracefully, absolute error <= 2^-36).
// Two values per call; ph / pl = the packed (x1, x0) pairs.  Domain: |x'| < 65520 (beyond it h = Inf, l = NaN: NaN out).
typedef _Float16 az_h2v __attribute__((ext_vector_type(2)));
typedef float az_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void az_split2h(float x0, float x1, unsigned& ph, unsigned& pl) {
  const az_f2v v = {x0 * AZ_F16X2_IN_SCALE, x1 * AZ_F16X2_IN_SCALE};
  const az_h2v h = __builtin_convertvector(v, az_h2v);
  // (x' - h) * 2^11 as ONE fused operation per value on the pre-scaled x (exact: the residual has at most 13 significant bits)
  const az_f2v t = {x0 * (AZ_F16X2_IN_SCALE * 2048.f), x1 * (AZ_F16X2_IN_SCALE * 2048.f)};
  const az_h2v l = {(_Float16)__builtin_fmaf((float)h.x, -2048.f, t.x), (_Float16)__builtin_fmaf((float)h.y, -2048.f, t.y)};
  ph = __builtin_bit_cast(unsigned, h);
  pl = __builtin_bit_cast(unsigned, l);
}

// The same split with a scale chosen at run time (AzConvArgs.in_absmax0: inputs whose magnitude is not bounded by construction):
// p and p2048 = 2048 p are wave-uniform powers of two.
__device__ __forceinline__ void az_split2h(float x0, float x1, float p, float p2048, unsigned& ph, unsigned& pl) {
  const az_f2v v = {x0 * p, x1 * p};
  const az_h2v h = __builtin_convertvector(v, az_h2v);
  const az_f2v t = {x0 * p2048, x1 * p2048};
  const az_h2v l = {(_Float16)__builtin_fmaf((float)h.x, -2048.f, t.x), (_Float16)__builtin_fmaf((float)h.y, -2048.f, t.y)};
  ph = __builtin_bit_cast(unsigned, h);
  pl = __builtin_bit_cast(unsigned, l);
}
// The activation scale of an f16x2 launch: AZ_F16X2_IN_SCALE, or -- with the absmax slots of its sources (az_absmax_f32) -- the
// power of two that puts gain * max |x| into [2^13, 2^14) (gain = 4 for the Winograd form: B^T d B sums four pixels), exponent
// clamped to +-60; a non-finite maximum keeps the fixed scale (the offending elements then poison only the outputs they reach).
// Wave-uniform (every lane reads the same 256 slots); the result sits in a scalar register.
__device__ __forceinline__ float az_f16x2_in_scale(const float* s0, const float* s1, float gain, int lane) {
  if (s0 == nullptr) return AZ_F16X2_IN_SCALE;
  float4 v = *reinterpret_cast<const float4*>(s0 + 4 * lane);
  float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
  if (s1 != nullptr) {
    v = *reinterpret_cast<const float4*>(s1 + 4 * lane);
    m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  m = az_wave_max(m) * gain;
  const unsigned mb = (unsigned)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(unsigned, m));
  int e = (int)((mb >> 23) & 255u) - 127;  // floor(log2 m) for normal m
  if (((mb >> 23) & 255u) == 255u) return AZ_F16X2_IN_SCALE;  // Inf / NaN
  int pe = 13 - e;
  pe = pe > 60 ? 60 : (pe < -60 ? -60 : pe);
  return __builtin_bit_cast(float, (unsigned)(127 + pe) << 23);
}

// The OTHER operand of an f16x2 product when it is formed inside a kernel (the keys and the probabilities of the attention kernel;
// packed weights get the same three pieces from layout.hip): x' = x * scale as ph = fp16(x'), pl = fp16(x' - ph) (the residual,
// unscaled) and phs = ph / 2^11, the factor of the partner's scaled low piece.  `scale` puts the operand's largest magnitudes near
// 2^10 .. 2^14, so that the unscaled residual stays a normal half for everything within ~2^-14 of them.
__device__ __forceinline__ void az_split_w2h(float x0, float x1, float scale, unsigned& ph, unsigned& pl, unsigned& phs) {
  const az_f2v v = {x0 * scale, x1 * scale};
  const az_h2v h = __builtin_convertvector(v, az_h2v);
  const az_f2v r = v - __builtin_convertvector(h, az_f2v);
  const az_h2v l = __builtin_convertvector(r, az_h2v);
  const az_h2v hs = h * (_Float16)0.00048828125f;
  ph = __builtin_bit_cast(unsigned, h);
  pl = __builtin_bit_cast(unsigned, l);
  phs = __builtin_bit_cast(unsigned, hs);
}

// Streaming accesses of the pure HBM streams (the transition kernels: every byte read once and written once per launch):
// non-temporal 16-byte loads / stores (global_load_dwordx4 ... nt).  Measured on the 96 Mi-element transition
// (profiles/r05_stream_nt_ab.txt): 12 B/element form 5.71 -> 6.14 TB/s, the 16 B form 5.46 -> 5.78, the 20 B form 5.40 -> 5.89;
// loads only or stores only give a third of it each.  The lines of an nt store still stay in the XCD's L2, so the consumer of
// a small latent (the stem convolution right behind the step) loses nothing.  NOT used by az_affine_act_f32: there the
// consumer re-reads a tensor that partly survives in the Infinity Cache, and the hint measured a loss inside an ADM step.
typedef float az_f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 az_ld_stream(const float* p) {
  const az_f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const az_f32x4_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void az_st_stream(float* p, float4 o) {
  const az_f32x4_t v = {o.x, o.y, o.z, o.w};
  __builtin_nontemporal_store(v, reinterpret_cast<az_f32x4_t*>(p));
}

// Element type of a launch's DESTINATION and RESIDUAL tensors (AzConvArgs.dst_dtype; modules cast to half precision keep their
// activations in HBM in the module's type, azula/denoise.py:314-320): IO = 0 fp32, 1 bfloat16, 2 IEEE half.  `base` is the
// tensor's address whatever the type, `idx` an ELEMENT index (a multiple of 4: 8-byte accesses for the 2-byte types); values are
// rounded to nearest even on the way out (what torch's cast does), arithmetic stays fp32.  IO is a compile-time parameter: the
// fp32 kernels instantiate IO = 0 only and compile to exactly the code they had.
typedef __bf16 az_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 az_f16x4 __attribute__((ext_vector_type(4)));
typedef float az_f32x4v __attribute__((ext_vector_type(4)));
template <int IO>
__device__ __forceinline__ float4 ld4_io(const float* base, int64_t idx) {
  if constexpr (IO == 0) {
    return *reinterpret_cast<const float4*>(base + idx);
  } else if constexpr (IO == 1) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base) + idx * 2);
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xFFFF0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xFFFF0000u));
  } else {
    const az_f16x4 h = *reinterpret_cast<const az_f16x4*>(reinterpret_cast<const char*>(base) + idx * 2);
    const az_f32x4v f = __builtin_convertvector(h, az_f32x4v);
    return make_float4(f.x, f.y, f.z, f.w);
  }
}
template <int IO>
__device__ __forceinline__ float4 round_io(float4 v) {  // the value the destination will hold (GroupNorm moments are taken of THAT)
  if constexpr (IO == 0) return v;
  const az_f32x4v f = {v.x, v.y, v.z, v.w};
  az_f32x4v r;
  if constexpr (IO == 1) r = __builtin_convertvector(__builtin_convertvector(f, az_bf16x4), az_f32x4v);
  else r = __builtin_convertvector(__builtin_convertvector(f, az_f16x4), az_f32x4v);
  return make_float4(r.x, r.y, r.z, r.w);
}
template <int IO>
__device__ __forceinline__ void st4_io(float* base, int64_t idx, float4 v, bool stream = false) {
  if constexpr (IO == 0) {
    if (stream) az_st_stream(base + idx, v);
    else *reinterpret_cast<float4*>(base + idx) = v;
  } else {
    const az_f32x4v f = {v.x, v.y, v.z, v.w};
    char* d = reinterpret_cast<char*>(base) + idx * 2;
    if constexpr (IO == 1) *reinterpret_cast<az_bf16x4*>(d) = __builtin_convertvector(f, az_bf16x4);
    else *reinterpret_cast<az_f16x4*>(d) = __builtin_convertvector(f, az_f16x4);
  }
}
template <int IO>
__device__ __forceinline__ void st2_io(float* base, int64_t idx, float2 v) {  // (the SwiGLU epilogue's half-width rows)
  if constexpr (IO == 0) {
    *reinterpret_cast<float2*>(base + idx) = v;
  } else {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v f = {v.x, v.y};
    char* d = reinterpret_cast<char*>(base) + idx * 2;
    if constexpr (IO == 1) {
      typedef __bf16 b2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<b2*>(d) = __builtin_convertvector(f, b2);
    } else {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<h2*>(d) = __builtin_convertvector(f, h2);
    }
  }
}

// Memory-bound launches: cap the grid at 256 CUs x 8 blocks and grid-stride the rest.
static inline int az_stream_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
