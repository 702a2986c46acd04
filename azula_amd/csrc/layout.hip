// Layout kernels: planar NCHW <-> channel-padded NHWC (the backbone-internal layout), and
// the one-off weight repack for the implicit-GEMM convolution.  HBM-bound, LDS-free: reads are
// coalesced along the pixel axis, writes are 16-byte channel vectors.
#include "common.h"

namespace {

// One thread per (b, pixel): reads C planes (coalesced across the wave), writes cs channels.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                           const float* __restrict__ scale, int64_t B, int64_t C,
                                                           int64_t HW, int64_t cs) {
  const float s = scale ? *scale : 1.0f;
  const int64_t total = B * HW;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = p / HW, i = p - b * HW;
    for (int64_t c0 = 0; c0 < cs; c0 += 4) {
      float v[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int64_t c = c0 + cc;
        v[cc] = c < C ? az_mul(s, src[(b * C + c) * HW + i]) : 0.f;
      }
      *reinterpret_cast<float4*>(dst + p * cs + c0) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                           int64_t B, int64_t C, int64_t HW, int64_t cs) {
  const int64_t total = B * HW;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = p / HW, i = p - b * HW;
    for (int64_t c0 = 0; c0 < C; c0 += 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + p * cs + c0);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if (c0 + cc < C) dst[(b * C + c0 + cc) * HW + i] = vv[cc];
    }
  }
}

// dst[tap][co][ci_packed] <- src[co][ci][tap]; zero padding everywhere else.
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                               int cout, int cin, int taps, int cout_s, int cin0,
                                                               int c0s, int cin_s) {
  const int64_t total = (int64_t)taps * cout_s * cin_s;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int cip = (int)(e % cin_s);
    const int co = (int)((e / cin_s) % cout_s);
    const int tap = (int)(e / ((int64_t)cin_s * cout_s));
    int ci = -1;
    if (cip < c0s) {
      if (cip < cin0) ci = cip;
    } else {
      const int r = cip - c0s;
      if (r < cin - cin0) ci = cin0 + r;
    }
    float v = 0.f;
    if (ci >= 0 && co < cout) v = src[((int64_t)co * cin + ci) * taps + tap];
    dst[e] = v;
  }
}

// Two-byte packing of the same [tap][co][ci_packed] layout for the bf16 / f16 MFMA kernel (round to nearest even).
template <bool F16>
__global__ __launch_bounds__(256) void pack_conv_weight_half_kernel(unsigned short* __restrict__ dst,
                                                                    const float* __restrict__ src, int cout, int cin,
                                                                    int taps, int cout_s, int cin0, int c0s, int cin_s) {
  const int64_t total = (int64_t)taps * cout_s * cin_s;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int cip = (int)(e % cin_s);
    const int co = (int)((e / cin_s) % cout_s);
    const int tap = (int)(e / ((int64_t)cin_s * cout_s));
    int ci = -1;
    if (cip < c0s) {
      if (cip < cin0) ci = cip;
    } else {
      const int r = cip - c0s;
      if (r < cin - cin0) ci = cin0 + r;
    }
    float v = 0.f;
    if (ci >= 0 && co < cout) v = src[((int64_t)co * cin + ci) * taps + tap];
    if (F16) {
      const _Float16 h = (_Float16)v;
      dst[e] = __builtin_bit_cast(unsigned short, h);
    } else {
      const __bf16 h = (__bf16)v;
      dst[e] = __builtin_bit_cast(unsigned short, h);
    }
  }
}

// Three-piece bf16 split of the same [tap][co][ci_packed] layout for the fp32-on-bf16-pipe kernel (conv_igemm_x3_kernel):
// dst = [piece][tap][cout_s][cin_s]; w = w1 + w2 + w3 exactly (8-bit slices of the significand, by truncation).
// The three 2-byte pieces of a packed weight.  H2 = false: v = p0 + p1 + p2 exactly (bf16, truncated 8-bit slices).  H2 = true
// ("f16x2", include/azula_amd.h): v' = v * w_scale as IEEE half pieces p0 = wh = fp16(v'), p1 = wl = fp16(v' - wh) (the residual,
// unscaled: subnormal for |v'| < 2^-3, i.e. below 2^-18 of the largest weight -- the matrix pipe honours subnormals), and
// p2 = wh / 2^11, the factor of the activation's scaled low piece.
template <bool H2>
__device__ __forceinline__ void weight_pieces(float v, float w_scale, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
  if constexpr (H2) {
    const float vs = v * w_scale;
    const _Float16 h = (_Float16)vs;
    const _Float16 l = (_Float16)(vs - (float)h);
    const _Float16 hs = (_Float16)((float)h * 0.00048828125f);
    p0 = __builtin_bit_cast(unsigned short, h);
    p1 = __builtin_bit_cast(unsigned short, l);
    p2 = __builtin_bit_cast(unsigned short, hs);
  } else {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const float r1 = v - __builtin_bit_cast(float, u & 0xFFFF0000u);
    const unsigned u1 = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    p0 = (unsigned short)(u >> 16);
    p1 = (unsigned short)(u1 >> 16);
    p2 = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
  }
}

template <bool H2>
__global__ __launch_bounds__(256) void pack_conv_weight_x3_kernel(unsigned short* __restrict__ dst,
                                                                  const float* __restrict__ src, int cout, int cin,
                                                                  int taps, int cout_s, int cin0, int c0s, int cin_s, float w_scale) {
  const int64_t total = (int64_t)taps * cout_s * cin_s;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int cip = (int)(e % cin_s);
    const int co = (int)((e / cin_s) % cout_s);
    const int tap = (int)(e / ((int64_t)cin_s * cout_s));
    int ci = -1;
    if (cip < c0s) {
      if (cip < cin0) ci = cip;
    } else {
      const int r = cip - c0s;
      if (r < cin - cin0) ci = cin0 + r;
    }
    float v = 0.f;
    if (ci >= 0 && co < cout) v = src[((int64_t)co * cin + ci) * taps + tap];
    weight_pieces<H2>(v, w_scale, dst[e], dst[total + e], dst[2 * total + e]);
  }
}

// Winograd F(2x2,3x3) filter transform U = G g G^T (accumulated in fp64, rounded once) written
// directly in the layout conv_winograd_kernel streams: [8-cin chunk][64-cout block][16][64][8].
__global__ __launch_bounds__(256) void winograd_filter_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                              int cout, int cin, int cin0, int nk0, int nk,
                                                              int cblocks) {
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const int64_t total = (int64_t)nk * cblocks * 16 * 64 * 8;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e & 7);
    const int col = (int)((e >> 3) & 63);
    const int f = (int)((e >> 9) & 15);
    const int64_t r = e >> 13;
    const int cbk = (int)(r % cblocks);
    const int kt = (int)(r / cblocks);
    const int co = cbk * 64 + col;
    int ci = -1;
    if (kt < nk0) {
      const int pc = kt * 8 + k;
      if (pc < cin0) ci = pc;
    } else {
      const int pc = (kt - nk0) * 8 + k;
      if (pc < cin - cin0) ci = cin0 + pc;
    }
    float v = 0.f;
    if (co < cout && ci >= 0) {
      const float* g = src + ((int64_t)co * cin + ci) * 9;
      const int xi = f >> 2, nu = f & 3;
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += G[xi][a] * (double)g[a * 3 + b] * G[nu][b];
      v = (float)acc;
    }
    dst[e] = v;
  }
}

// The same transform for az_conv2d_winograd_x3_f32 (wino_x3.hip): U rounded to fp32 exactly as above, then split into three bf16
// pieces (u = u1 + u2 + u3 exactly, 8-bit slices of the significand by truncation) and stored in the order the kernel's waves load
// it, as v_mfma_f32_32x32x16_bf16 A fragments: [16-cin step][64-cout block][wave w = 0..7][f][cout half][piece][lane][8 bf16], where
// wave w owns the frequencies (xi = w & 3, nu = (w >> 2) + 2 f) of all 64 couts and lane (l31 = lane & 31, h = lane >> 5) holds
// U[xi, nu][cout 32 half + l31][channels 8 h .. 8 h + 7] -- twelve contiguous 1 KB pieces per wave and step; the filter never
// passes through LDS.
template <bool H2>  // (H2: the f16x2 pieces of U * w_scale, see weight_pieces)
__global__ __launch_bounds__(256) void winograd_filter_x3_kernel(unsigned short* __restrict__ dst, const float* __restrict__ src,
                                                                 int cout, int cin, int cin0, int nk0, int nk, int cblocks, float w_scale) {
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const int64_t total = (int64_t)nk * cblocks * 8 * 4 * 64 * 8;  // values (three pieces each)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(e & 7);
    const int ln = (int)((e >> 3) & 63);
    const int ch = (int)((e >> 9) & 1);
    const int f = (int)((e >> 10) & 1);
    const int w = (int)((e >> 11) & 7);
    const int64_t r = e >> 14;
    const int cbk = (int)(r % cblocks);
    const int kt = (int)(r / cblocks);
    const int co = cbk * 64 + ch * 32 + (ln & 31);
    const int k = 8 * (ln >> 5) + k8;
    int ci = -1;
    if (kt < nk0) {
      const int pc = kt * 16 + k;
      if (pc < cin0) ci = pc;
    } else {
      const int pc = (kt - nk0) * 16 + k;
      if (pc < cin - cin0) ci = cin0 + pc;
    }
    float v = 0.f;
    if (co < cout && ci >= 0) {
      const float* g = src + ((int64_t)co * cin + ci) * 9;
      const int xi = w & 3, nu = (w >> 2) + 2 * f;
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += G[xi][a] * (double)g[a * 3 + b] * G[nu][b];
      v = (float)acc;
    }
    unsigned short* d = dst + (((((r * 8 + w) * 2 + f) * 2 + ch) * 3) * 64 + ln) * 8 + k8;  // piece 0; pieces are 512 elements apart
    weight_pieces<H2>(v, w_scale, d[0], d[512], d[1024]);
  }
}

// Winograd F(4x4,3x3) filter transform U = G g G^T (6x6 per filter, interpolation points 0, +-1, +-2, inf;
// fp64 accumulate, rounded once) in the layout conv_winograd4_kernel streams:
// [4-cin chunk][64-cout block][36][64][4].
__global__ __launch_bounds__(256) void winograd4_filter_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                               int cout, int cin, int cin0, int c0s, int nk,
                                                               int cblocks) {
  const double G[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  const int64_t total = (int64_t)nk * cblocks * 36 * 64 * 4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e & 3);
    const int col = (int)((e >> 2) & 63);
    const int64_t r0 = e >> 8;
    const int f = (int)(r0 % 36);
    const int64_t r = r0 / 36;
    const int cbk = (int)(r % cblocks);
    const int kt = (int)(r / cblocks);
    const int co = cbk * 64 + col;
    const int pc = kt * 4 + k;  // position in the [source 0 | source 1] channel concatenation
    int ci = -1;
    if (pc < c0s) {
      if (pc < cin0) ci = pc;
    } else if (pc - c0s < cin - cin0) {
      ci = cin0 + pc - c0s;
    }
    float v = 0.f;
    if (co < cout && ci >= 0) {
      const float* g = src + ((int64_t)co * cin + ci) * 9;
      const int xi = f / 6, nu = f - xi * 6;
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += G[xi][a] * (double)g[a * 3 + b] * G[nu][b];
      v = (float)acc;
    }
    dst[e] = v;
  }
}


// Nearest upsampling by an arbitrary integer factor per axis, cropped to (Hout, Wout) (the reference's
// Upsample(scale_factor = stride) followed by narrow, azula/nn/unet.py:186,250-252): dst[b, y, x, :] = src[b, y / sh, x / sw, :].
// Power-of-two factors never come here (they are a right shift inside the merge convolution's gather); this pass exists
// for the other strides (3, 5, 6 ...).  NHWC, one float4 per thread.
// The source index is ATen's: min(floor(dst * float(1.0 / scale)), in - 1) in fp32 (upsample_nearest with a given scale factor).
__global__ __launch_bounds__(256) void upsample_nearest_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t total4,
                                                               int Hin, int Win, int q, float inv_h, float inv_w, int Hout, int Wout) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % q);
    int64_t r = i / q;
    const int x = (int)(r % Wout);
    r /= Wout;
    const int y = (int)(r % Hout);
    const int64_t b = r / Hout;
    const int ys = min((int)floorf(__fmul_rn((float)y, inv_h)), Hin - 1);
    const int xs = min((int)floorf(__fmul_rn((float)x, inv_w)), Win - 1);
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[((b * Hin + ys) * Win + xs) * q + c4];
  }
}

}  // namespace

extern "C" {

int az_nchw_to_nhwc_f32(float* dst, const float* src, const float* scale_dev, int64_t B, int64_t C, int64_t HW,
                        int64_t cs, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && C > 0 && HW > 0 && cs >= C && cs % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(dst), AZ_E_ALIGN);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(az_stream_grid(B * HW, 256)), dim3(256), 0, az_s(stream), dst, src,
                     scale_dev, B, C, HW, cs);
  return az_launch_status();
}

int az_upsample_nearest_f32(float* dst, const float* src, int64_t B, int64_t Hin, int64_t Win, int64_t cs, int32_t sh,
                            int32_t sw, int64_t Hout, int64_t Wout, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && Hin > 0 && Win > 0 && cs > 0 && cs % 4 == 0 && sh >= 1 && sw >= 1, AZ_E_SHAPE);
  AZ_REQUIRE(Hout > 0 && Wout > 0 && (Hout + sh - 1) / sh <= Hin && (Wout + sw - 1) / sw <= Win, AZ_E_SHAPE);
  AZ_REQUIRE(Hin < (1 << 30) && Win < (1 << 30) && Hout < (1 << 30) && Wout < (1 << 30), AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(dst) && AZ_ALIGNED16(src), AZ_E_ALIGN);
  const int64_t total4 = B * Hout * Wout * (cs / 4);
  hipLaunchKernelGGL(upsample_nearest_kernel, dim3(az_stream_grid(total4, 256)), dim3(256), 0, az_s(stream), dst, src, total4,
                     (int)Hin, (int)Win, (int)(cs / 4), (float)(1.0 / sh), (float)(1.0 / sw), (int)Hout, (int)Wout);
  return az_launch_status();
}

int az_nhwc_to_nchw_f32(float* dst, const float* src, int64_t B, int64_t C, int64_t HW, int64_t cs,
                        az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(B > 0 && C > 0 && HW > 0 && cs >= C && cs % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(src), AZ_E_ALIGN);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(az_stream_grid(B * HW, 256)), dim3(256), 0, az_s(stream), dst, src,
                     B, C, HW, cs);
  return az_launch_status();
}

int az_pack_conv_weight_f32(float* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                            int32_t cin0, int32_t c0s, int32_t cin_s, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && ks > 0 && cout_s >= cout && cout_s % 4 == 0 && cin_s % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(cin0 >= 0 && cin0 <= cin && c0s >= cin0 && cin_s >= c0s + (cin - cin0), AZ_E_SHAPE);
  const int64_t total = (int64_t)ks * ks * cout_s * cin_s;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream), dst, src,
                     cout, cin, ks * ks, cout_s, cin0, c0s, cin_s);
  return az_launch_status();
}

int az_pack_conv_weight_half_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                                 int32_t cin0, int32_t c0s, int32_t cin_s, int32_t f16, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && ks > 0 && cout_s >= cout && cout_s % 4 == 0 && cin_s % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(cin0 >= 0 && cin0 <= cin && c0s >= cin0 && cin_s >= c0s + (cin - cin0), AZ_E_SHAPE);
  const int64_t total = (int64_t)ks * ks * cout_s * cin_s;
  if (f16)
    hipLaunchKernelGGL(pack_conv_weight_half_kernel<true>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream),
                       (unsigned short*)dst, src, cout, cin, ks * ks, cout_s, cin0, c0s, cin_s);
  else
    hipLaunchKernelGGL(pack_conv_weight_half_kernel<false>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream),
                       (unsigned short*)dst, src, cout, cin, ks * ks, cout_s, cin0, c0s, cin_s);
  return az_launch_status();
}

int az_pack_conv_weight_x3_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                               int32_t cin0, int32_t c0s, int32_t cin_s, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && ks > 0 && cout_s >= cout && cout_s % 4 == 0 && cin_s % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE(cin0 >= 0 && cin0 <= cin && c0s >= cin0 && cin_s >= c0s + (cin - cin0), AZ_E_SHAPE);
  const int64_t total = (int64_t)ks * ks * cout_s * cin_s;
  hipLaunchKernelGGL(pack_conv_weight_x3_kernel<false>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream),
                     (unsigned short*)dst, src, cout, cin, ks * ks, cout_s, cin0, c0s, cin_s, 1.f);
  return az_launch_status();
}

static bool az_pow2(float v) {  // a finite positive power of two
  int e;
  return v > 0.f && v < 3.0e38f && frexpf(v, &e) == 0.5f;
}

float az_f16x2_weight_scale(float amax, int32_t winograd) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(winograd ? amax * 2.25f : amax, &e);  // value in [2^(e-1), 2^e)
  int k = 14 - e;                                     // value * 2^k in [2^13, 2^14)
  if (k > 100) k = 100;
  if (k < -100) k = -100;
  return ldexpf(1.f, k);
}

int az_pack_conv_weight_f16x2_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                                  int32_t cin0, int32_t c0s, int32_t cin_s, float w_scale, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && ks > 0 && cout_s >= cout && cout_s % 4 == 0 && cin_s % 4 == 0 && az_pow2(w_scale), AZ_E_SHAPE);
  AZ_REQUIRE(cin0 >= 0 && cin0 <= cin && c0s >= cin0 && cin_s >= c0s + (cin - cin0), AZ_E_SHAPE);
  const int64_t total = (int64_t)ks * ks * cout_s * cin_s;
  hipLaunchKernelGGL(pack_conv_weight_x3_kernel<true>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream),
                     (unsigned short*)dst, src, cout, cin, ks * ks, cout_s, cin0, c0s, cin_s, w_scale);
  return az_launch_status();
}

int az_winograd_pack_filter_f32(float* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t nk0,
                                int32_t nk, int32_t cblocks, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && cin0 >= 0 && cin0 <= cin && nk0 * 8 >= cin0 && (nk - nk0) * 8 >= cin - cin0 &&
                 cblocks * 64 >= cout,
             AZ_E_SHAPE);
  const int64_t total = (int64_t)nk * cblocks * 16 * 64 * 8;
  hipLaunchKernelGGL(winograd_filter_kernel, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream), dst, src,
                     cout, cin, cin0, nk0, nk, cblocks);
  return az_launch_status();
}

int az_winograd_pack_filter_x3_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t nk0,
                                   int32_t nk, int32_t cblocks, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && cin0 >= 0 && cin0 <= cin && nk0 * 16 >= cin0 && (nk - nk0) * 16 >= cin - cin0 &&
                 cblocks * 64 >= cout,
             AZ_E_SHAPE);
  const int64_t total = (int64_t)nk * cblocks * 8 * 4 * 64 * 8;
  hipLaunchKernelGGL(winograd_filter_x3_kernel<false>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream),
                     (unsigned short*)dst, src, cout, cin, cin0, nk0, nk, cblocks, 1.f);
  return az_launch_status();
}

int az_winograd_pack_filter_f16x2_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t nk0,
                                      int32_t nk, int32_t cblocks, float w_scale, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && cin0 >= 0 && cin0 <= cin && nk0 * 16 >= cin0 && (nk - nk0) * 16 >= cin - cin0 &&
                 cblocks * 64 >= cout && az_pow2(w_scale),
             AZ_E_SHAPE);
  const int64_t total = (int64_t)nk * cblocks * 8 * 4 * 64 * 8;
  hipLaunchKernelGGL(winograd_filter_x3_kernel<true>, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream),
                     (unsigned short*)dst, src, cout, cin, cin0, nk0, nk, cblocks, w_scale);
  return az_launch_status();
}

int az_winograd4_pack_filter_f32(float* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t c0s,
                                 int32_t nk, int32_t cblocks, az_stream_t stream) {
  AZ_REQUIRE(dst && src, AZ_E_NULL);
  AZ_REQUIRE(cout > 0 && cin > 0 && cin0 >= 0 && cin0 <= cin && c0s >= cin0 && c0s % 4 == 0 &&
                 nk * 4 >= c0s + (cin - cin0) && cblocks * 64 >= cout,
             AZ_E_SHAPE);
  const int64_t total = (int64_t)nk * cblocks * 36 * 64 * 4;
  hipLaunchKernelGGL(winograd4_filter_kernel, dim3(az_stream_grid(total, 256)), dim3(256), 0, az_s(stream), dst, src,
                     cout, cin, cin0, c0s, nk, cblocks);
  return az_launch_status();
}

}  // extern "C"
