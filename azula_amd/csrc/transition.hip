// K1: fused Karras/ADM post-conditioning + (CFG combine) + DDPM/DDIM transition + next-step
// pre-scale, plus the tiny per-step scalar plumbing kernels.  HBM-bound: one pass over the
// latent, 16-byte accesses, grid capped at 2048 blocks with a grid-stride loop.
//
// Reference arithmetic replaced (see include/azula_amd.h): azula/denoise.py:317,322,
// azula/plugins/adm/__init__.py:126-134, azula/guidance/cfg.py:63-65, azula/sample.py:257-259.
#include "common.h"

namespace {

struct Coef {
  float c_skip, c_out, alpha_t, alpha_s, k_x, k_eps, c_in_next, lo, hi, g;
};

__device__ __forceinline__ Coef load_coef(const AzStepCoef* c) {
  Coef k;
  k.c_skip = c->c_skip;
  k.c_out = c->c_out;
  k.alpha_t = c->alpha_t;
  k.alpha_s = c->alpha_s;
  k.k_x = c->k_x;
  k.k_eps = c->k_eps;
  k.c_in_next = c->c_in_next;
  k.lo = c->clip_lo;
  k.hi = c->clip_hi;
  k.g = c->guidance;
  return k;
}

// torch.clip semantics: NaN propagates (fminf / fmaxf alone would turn a NaN mean into the bound and hide a
// diverging run that the reference shows as NaN -- azula/plugins/adm/__init__.py:131-134).
__device__ __forceinline__ float clip_nan(float m, float lo, float hi) {
  const float c = fminf(fmaxf(m, lo), hi);
  return m != m ? m : c;
}

// Posterior mean for one element, reference association order, every op rounded on its own.
template <bool CFG>
__device__ __forceinline__ float post_mean(const Coef& k, float x, float f, float fn) {
  float m = az_add(az_mul(k.c_skip, x), az_mul(k.c_out, f));
  m = clip_nan(m, k.lo, k.hi);
  if (CFG) {
    float mn = az_add(az_mul(k.c_skip, x), az_mul(k.c_out, fn));
    mn = clip_nan(mn, k.lo, k.hi);
    m = az_add(m, az_mul(k.g, az_sub(m, mn)));
  }
  return m;
}

template <bool EPS>
__device__ __forceinline__ float step_x(const Coef& k, float x, float m, float e) {
  float xs = az_mul(k.alpha_s, m);
  xs = az_add(xs, az_mul(k.k_x, az_sub(x, az_mul(k.alpha_t, m))));
  if (EPS) xs = az_add(xs, az_mul(k.k_eps, e));
  return xs;
}

constexpr int TF_UN = 4;  // float4 per thread and stream in flight in the flat kernel

// ---- flat elementwise form: every tensor has the same (n,) layout --------------------------
template <bool CFG, bool EPS, bool XIN, bool MEAN>
__global__ __launch_bounds__(256) void transition_flat_kernel(const float* __restrict__ x_t,
                                                              const float* __restrict__ F,
                                                              const float* __restrict__ Fn,
                                                              const float* __restrict__ eps, float* x_s,
                                                              float* __restrict__ xin, float* __restrict__ mean_out,
                                                              int64_t n4, int64_t n, const AzStepCoef* coef,
                                                              int64_t xstep, int64_t fstep) {
  // blockIdx.y = image, only when F carries more planar channels per image than x (ADM: 6, the first 3 are read): every
  // image is then a flat problem of its own (xstep / fstep elements apart); otherwise gridDim.y = 1 and the steps are 0
  x_t += blockIdx.y * xstep;
  x_s += blockIdx.y * xstep;
  F += blockIdx.y * fstep;
  if (CFG) Fn += blockIdx.y * fstep;
  if (EPS) eps += blockIdx.y * xstep;
  if (XIN) xin += blockIdx.y * xstep;
  if (MEAN) mean_out += blockIdx.y * xstep;
  const Coef k = load_coef(coef);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto one = [&](int64_t i, float4 xv, float4 fv, float4 nv, float4 ev) {
    float4 m, o;
    m.x = post_mean<CFG>(k, xv.x, fv.x, nv.x);
    m.y = post_mean<CFG>(k, xv.y, fv.y, nv.y);
    m.z = post_mean<CFG>(k, xv.z, fv.z, nv.z);
    m.w = post_mean<CFG>(k, xv.w, fv.w, nv.w);
    o.x = step_x<EPS>(k, xv.x, m.x, ev.x);
    o.y = step_x<EPS>(k, xv.y, m.y, ev.y);
    o.z = step_x<EPS>(k, xv.z, m.z, ev.z);
    o.w = step_x<EPS>(k, xv.w, m.w, ev.w);
    az_st_stream(x_s + 4 * i, o);
    if (MEAN) az_st_stream(mean_out + 4 * i, m);
    if (XIN) {
      float4 q;
      q.x = az_mul(k.c_in_next, o.x);
      q.y = az_mul(k.c_in_next, o.y);
      q.z = az_mul(k.c_in_next, o.z);
      q.w = az_mul(k.c_in_next, o.w);
      az_st_stream(xin + 4 * i, q);
    }
  };
  // x_s may alias x_t (in-place step), so the compiler cannot hoist the next element's loads above this element's
  // store: the loop is unrolled by hand with all loads of UN elements issued before the first store (every element is
  // read and written by the same thread only, so this is safe in place) -- 2-4 x UN KB in flight per wave.
  constexpr int UN = TF_UN;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // a workgroup owns UN consecutive 4 KB pieces of every stream per iteration (16 KB contiguous per array)
  const int64_t span = (int64_t)UN * blockDim.x;
  int64_t base = (int64_t)blockIdx.x * span;
  for (; base + span <= n4; base += (int64_t)gridDim.x * span) {
    const int64_t i = base + threadIdx.x;
    float4 xv[UN], fv[UN], nv[UN], ev[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t e = 4 * (i + u * blockDim.x);  // (streams read once: non-temporal, see common.h)
      xv[u] = az_ld_stream(x_t + e);
      fv[u] = az_ld_stream(F + e);
      nv[u] = CFG ? az_ld_stream(Fn + e) : z4;
      ev[u] = EPS ? az_ld_stream(eps + e) : z4;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) one(i + u * blockDim.x, xv[u], fv[u], nv[u], ev[u]);
  }
  // remainder (fewer than gridDim.x * span float4): plain grid-stride over what is left
  {
    const int64_t done = n4 / span * span;
    for (int64_t i = done + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
      one(i, reinterpret_cast<const float4*>(x_t)[i], reinterpret_cast<const float4*>(F)[i],
          CFG ? reinterpret_cast<const float4*>(Fn)[i] : z4, EPS ? reinterpret_cast<const float4*>(eps)[i] : z4);
  }
  // scalar tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float x = x_t[i];
      const float m = post_mean<CFG>(k, x, F[i], CFG ? Fn[i] : 0.f);
      const float o = step_x<EPS>(k, x, m, EPS ? eps[i] : 0.f);
      x_s[i] = o;
      if (MEAN) mean_out[i] = m;
      if (XIN) xin[i] = az_mul(k.c_in_next, o);
    }
  }
}

// ---- image form: x is (B, C, inner) planar; F may be planar with more channels (ADM: 6 of
// which 3 are read) or NHWC with channel stride fC; xin may be written NHWC with stride xs_c.
// One thread owns 4 consecutive pixels of one sample: all accesses are 16 B.  x_s may alias x_t (the captured loop
// steps in place), so the compiler cannot move a later channel's loads above an earlier channel's stores: every load
// of a 4-channel chunk (x, F, F_neg, eps: up to 16 x 16 B per thread) is issued BEFORE its first store, as in the flat
// kernel -- each element is read and written by the same thread only, so this is safe in place.
template <bool CFG, bool EPS, bool MEAN>
__global__ __launch_bounds__(256) void transition_image_kernel(AzTransitionArgs a, int64_t quads_per_sample) {
  const Coef k = load_coef(a.coef);
  const int64_t total = a.batch * quads_per_sample;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int C = (int)a.channels;
  const int64_t inner = a.inner;
  const int64_t fC = a.f_channels;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
    const int64_t b = q / quads_per_sample;
    const int64_t i0 = (q - b * quads_per_sample) * 4;
    for (int c0 = 0; c0 < C; c0 += 4) {
      const int nc = C - c0 < 4 ? C - c0 : 4;
      float4 xv[4], fv[4], nv[4], ev[4];  // [channel in chunk] x 4 pixels
      // ---- all loads of the chunk
      if (a.f_nhwc) {
        float4 fpix[4], npix[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t off = (b * inner + i0 + j) * fC + c0;
          fpix[j] = *reinterpret_cast<const float4*>(a.F + off);
          npix[j] = CFG ? *reinterpret_cast<const float4*>(a.F_neg + off) : z4;
        }
        // 4 x 4 transpose: pixel-major -> channel-major
        fv[0] = make_float4(fpix[0].x, fpix[1].x, fpix[2].x, fpix[3].x);
        fv[1] = make_float4(fpix[0].y, fpix[1].y, fpix[2].y, fpix[3].y);
        fv[2] = make_float4(fpix[0].z, fpix[1].z, fpix[2].z, fpix[3].z);
        fv[3] = make_float4(fpix[0].w, fpix[1].w, fpix[2].w, fpix[3].w);
        nv[0] = make_float4(npix[0].x, npix[1].x, npix[2].x, npix[3].x);
        nv[1] = make_float4(npix[0].y, npix[1].y, npix[2].y, npix[3].y);
        nv[2] = make_float4(npix[0].z, npix[1].z, npix[2].z, npix[3].z);
        nv[3] = make_float4(npix[0].w, npix[1].w, npix[2].w, npix[3].w);
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        xv[cc] = z4;
        ev[cc] = z4;
        if (!a.f_nhwc) fv[cc] = nv[cc] = z4;
        if (cc < nc) {
          const int64_t xo = (b * C + c0 + cc) * inner + i0;
          xv[cc] = *reinterpret_cast<const float4*>(a.x_t + xo);
          if (EPS) ev[cc] = *reinterpret_cast<const float4*>(a.eps + xo);
          if (!a.f_nhwc) {
            const int64_t fo = (b * fC + c0 + cc) * inner + i0;
            fv[cc] = *reinterpret_cast<const float4*>(a.F + fo);
            if (CFG) nv[cc] = *reinterpret_cast<const float4*>(a.F_neg + fo);
          }
        }
      }
      // ---- arithmetic + stores
      float4 ov[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        ov[cc] = z4;
        if (cc < nc) {
          const int64_t xo = (b * C + c0 + cc) * inner + i0;
          float4 m, o;
          m.x = post_mean<CFG>(k, xv[cc].x, fv[cc].x, nv[cc].x);
          m.y = post_mean<CFG>(k, xv[cc].y, fv[cc].y, nv[cc].y);
          m.z = post_mean<CFG>(k, xv[cc].z, fv[cc].z, nv[cc].z);
          m.w = post_mean<CFG>(k, xv[cc].w, fv[cc].w, nv[cc].w);
          o.x = step_x<EPS>(k, xv[cc].x, m.x, ev[cc].x);
          o.y = step_x<EPS>(k, xv[cc].y, m.y, ev[cc].y);
          o.z = step_x<EPS>(k, xv[cc].z, m.z, ev[cc].z);
          o.w = step_x<EPS>(k, xv[cc].w, m.w, ev[cc].w);
          *reinterpret_cast<float4*>(a.x_s + xo) = o;
          if (MEAN) *reinterpret_cast<float4*>(a.mean_out + xo) = m;
          ov[cc] = o;
        }
      }
      if (a.xin_next != nullptr) {
        if (a.nhwc_pad > 0) {
          const float4 p0 = make_float4(az_mul(k.c_in_next, ov[0].x), az_mul(k.c_in_next, ov[1].x),
                                        az_mul(k.c_in_next, ov[2].x), az_mul(k.c_in_next, ov[3].x));
          const float4 p1 = make_float4(az_mul(k.c_in_next, ov[0].y), az_mul(k.c_in_next, ov[1].y),
                                        az_mul(k.c_in_next, ov[2].y), az_mul(k.c_in_next, ov[3].y));
          const float4 p2 = make_float4(az_mul(k.c_in_next, ov[0].z), az_mul(k.c_in_next, ov[1].z),
                                        az_mul(k.c_in_next, ov[2].z), az_mul(k.c_in_next, ov[3].z));
          const float4 p3 = make_float4(az_mul(k.c_in_next, ov[0].w), az_mul(k.c_in_next, ov[1].w),
                                        az_mul(k.c_in_next, ov[2].w), az_mul(k.c_in_next, ov[3].w));
          float* dst = a.xin_next + (b * inner + i0) * a.nhwc_pad + c0;
          *reinterpret_cast<float4*>(dst) = p0;
          *reinterpret_cast<float4*>(dst + a.nhwc_pad) = p1;
          *reinterpret_cast<float4*>(dst + 2 * a.nhwc_pad) = p2;
          *reinterpret_cast<float4*>(dst + 3 * a.nhwc_pad) = p3;
        } else {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            if (cc >= nc) continue;
            float4 v;
            v.x = az_mul(k.c_in_next, ov[cc].x);
            v.y = az_mul(k.c_in_next, ov[cc].y);
            v.z = az_mul(k.c_in_next, ov[cc].z);
            v.w = az_mul(k.c_in_next, ov[cc].w);
            *reinterpret_cast<float4*>(a.xin_next + (b * C + c0 + cc) * inner + i0) = v;
          }
        }
      }
    }
    // zero the remaining pad channels of the NHWC pre-scaled output
    if (a.xin_next != nullptr && a.nhwc_pad > 0) {
      for (int c0 = (C + 3) & ~3; c0 < a.nhwc_pad; c0 += 4)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(a.xin_next + (b * inner + i0 + j) * a.nhwc_pad + c0) =
              make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// ---- image form, hot case (UNet / ADM latents): C <= 4 planar channels, pre-scaled NHWC output with a 4-float pixel
// stride.  Per workgroup pass: 256 threads x 4 consecutive pixels.  Planar streams are 16 B/lane as above; the NHWC
// stream is NOT written from the owning lane (lane stride 64 B: a wave store would touch 32 half-written 128-byte
// lines) -- the scaled outputs are transposed through 16 KB of LDS ([channel][pixel], conflict-free ds_write_b128 /
// ds_read_b32) so that every wave store writes 1 KB of consecutive pixels.  Measured at 96 Mi elements (DDIM eta=0,
// 16 B/element algorithmic): 3.97 TB/s with per-lane NHWC stores -> see DESIGN.md section 4 for the current number.
constexpr int TI_UN = 4;  // pixel quads per thread and pass in the image4 kernel (all their loads before the first store)

template <bool CFG, bool EPS, bool MEAN>
__global__ __launch_bounds__(256) void transition_image4_kernel(AzTransitionArgs a, int64_t quads_per_sample,
                                                                int64_t total) {
  constexpr int UN = TI_UN;
  __shared__ __attribute__((aligned(16))) float sm[4][1024 * UN];
  const Coef k = load_coef(a.coef);
  const int C = (int)a.channels;
  const int64_t inner = a.inner;
  const int64_t fC = a.f_channels;
  const int tid = threadIdx.x;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t q0 = (int64_t)blockIdx.x * (256 * UN); q0 < total; q0 += (int64_t)gridDim.x * (256 * UN)) {
    float4 xv[UN][4], fv[UN][4], nv[UN][4], ev[UN][4];
    int64_t xo[UN][4];
    bool valid[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t q = q0 + u * 256 + tid;
      valid[u] = q < total;
      const int64_t b = valid[u] ? q / quads_per_sample : 0;
      const int64_t i0 = valid[u] ? (q - b * quads_per_sample) * 4 : 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xv[u][c] = fv[u][c] = nv[u][c] = ev[u][c] = z4;
        xo[u][c] = (b * C + c) * inner + i0;
        if (valid[u] && c < C) {
          const int64_t fo = (b * fC + c) * inner + i0;
          xv[u][c] = az_ld_stream(a.x_t + xo[u][c]);
          fv[u][c] = az_ld_stream(a.F + fo);
          if (CFG) nv[u][c] = az_ld_stream(a.F_neg + fo);
          if (EPS) ev[u][c] = az_ld_stream(a.eps + xo[u][c]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4 sc = z4;
        if (valid[u] && c < C) {
          float4 m, o;
          m.x = post_mean<CFG>(k, xv[u][c].x, fv[u][c].x, nv[u][c].x);
          m.y = post_mean<CFG>(k, xv[u][c].y, fv[u][c].y, nv[u][c].y);
          m.z = post_mean<CFG>(k, xv[u][c].z, fv[u][c].z, nv[u][c].z);
          m.w = post_mean<CFG>(k, xv[u][c].w, fv[u][c].w, nv[u][c].w);
          o.x = step_x<EPS>(k, xv[u][c].x, m.x, ev[u][c].x);
          o.y = step_x<EPS>(k, xv[u][c].y, m.y, ev[u][c].y);
          o.z = step_x<EPS>(k, xv[u][c].z, m.z, ev[u][c].z);
          o.w = step_x<EPS>(k, xv[u][c].w, m.w, ev[u][c].w);
          az_st_stream(a.x_s + xo[u][c], o);
          if (MEAN) az_st_stream(a.mean_out + xo[u][c], m);
          sc = make_float4(az_mul(k.c_in_next, o.x), az_mul(k.c_in_next, o.y), az_mul(k.c_in_next, o.z),
                           az_mul(k.c_in_next, o.w));
        }
        *reinterpret_cast<float4*>(&sm[c][1024 * u + 4 * tid]) = sc;
      }
    __syncthreads();
    // pixel P = 4 q + j of the flattened (sample, pixel) index lives at xin + 4 P: contiguous across the workgroup
    float* dst = a.xin_next + q0 * 16;
    const int64_t npx = (total - q0) * 4;  // pixels left from q0 on
#pragma unroll
    for (int i = 0; i < 4 * UN; ++i) {
      const int p = 256 * i + tid;
      if (p < npx) az_st_stream(dst + 4 * (int64_t)p, make_float4(sm[0][p], sm[1][p], sm[2][p], sm[3][p]));
    }
    __syncthreads();
  }
}

__global__ void step_begin_kernel(AzStepCoef* cur, const AzStepCoef* table, int32_t* counter, int32_t n_steps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int32_t s = *counter;
    if (s < 0) s = 0;
    if (s >= n_steps) s = n_steps - 1;
    *cur = table[s];
    *counter = *counter + 1;
  }
}

// fp64 row of the step that az_step_begin has just started (the counter already points at the next one)
__global__ void step_row_f64_kernel(double* cur, const double* table, const int32_t* counter, int32_t n_rows, int32_t words) {
  int32_t s = *counter - 1;
  if (s < 0) s = 0;
  if (s >= n_rows) s = n_rows - 1;
  if ((int)threadIdx.x < words) cur[threadIdx.x] = table[(int64_t)s * words + threadIdx.x];
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                    const float* __restrict__ s, int64_t n4, int64_t n) {
  const float k = *s;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = az_mul(k, v.x);
    v.y = az_mul(k, v.y);
    v.z = az_mul(k, v.z);
    v.w = az_mul(k, v.w);
    reinterpret_cast<float4*>(y)[i] = v;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) y[i] = az_mul(k, x[i]);
}

__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ y, const float* __restrict__ a,
                                                    const float* __restrict__ x, const float* __restrict__ b,
                                                    const float* __restrict__ z, int64_t rows, int64_t inner,
                                                    int a_stride) {
  const int64_t n = rows * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = a_stride ? i / inner : 0;
    y[i] = az_add(az_mul(a[r * a_stride], x[i]), az_mul(b[r * a_stride], z[i]));
  }
}

// Linear multistep update (Adams-Bashforth family, reference azula/sample.py:519-546 and the
// v/zE/xE/RE variants): one pass computes the newest prediction pred = a x_t + b mean, stores it
// in its history slot and forms x_s = p x_t + sum_j w_j hist_j + w_new pred.  Reads 2 + n_hist
// streams and writes 2: the HBM minimum for the step.  `Vec` = float4 body / float tail.
struct MultistepPtrs {
  const float* h[AZ_MULTISTEP_MAX_HIST];
};

template <int NH>
__global__ __launch_bounds__(256) void multistep_kernel(float* x_s /* may alias x_t (in-place step) */,
                                                        float* __restrict__ pred, const float* x_t,
                                                        const float* __restrict__ mean,
                                                        MultistepPtrs hist, const float* __restrict__ coef,
                                                        int64_t n) {
  const float a = coef[0], b = coef[1], p = coef[2], wn = coef[3];
  float w[NH > 0 ? NH : 1];
#pragma unroll
  for (int j = 0; j < NH; ++j) w[j] = coef[4 + j];
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 xv = reinterpret_cast<const float4*>(x_t)[i];
    const float4 mv = reinterpret_cast<const float4*>(mean)[i];
    float4 hv[NH > 0 ? NH : 1];
#pragma unroll
    for (int j = 0; j < NH; ++j) hv[j] = reinterpret_cast<const float4*>(hist.h[j])[i];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ms[4] = {mv.x, mv.y, mv.z, mv.w};
    float pr[4], out[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      pr[c] = az_add(az_mul(a, xs[c]), az_mul(b, ms[c]));
      float acc = az_mul(p, xs[c]);
#pragma unroll
      for (int j = 0; j < NH; ++j) acc = az_add(acc, az_mul(w[j], reinterpret_cast<const float*>(&hv[j])[c]));
      out[c] = az_add(acc, az_mul(wn, pr[c]));
    }
    reinterpret_cast<float4*>(pred)[i] = make_float4(pr[0], pr[1], pr[2], pr[3]);
    reinterpret_cast<float4*>(x_s)[i] = make_float4(out[0], out[1], out[2], out[3]);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float x = x_t[i];
      const float pr = az_add(az_mul(a, x), az_mul(b, mean[i]));
      float acc = az_mul(p, x);
#pragma unroll
      for (int j = 0; j < NH; ++j) acc = az_add(acc, az_mul(w[j], hist.h[j][i]));
      pred[i] = pr;
      x_s[i] = az_add(acc, az_mul(wn, pr));
    }
}

template <int NH>
void launch_multistep(const AzMultistepArgs* a, hipStream_t s) {
  MultistepPtrs hp;
  for (int j = 0; j < AZ_MULTISTEP_MAX_HIST; ++j) hp.h[j] = j < NH ? a->hist[j] : nullptr;
  hipLaunchKernelGGL(multistep_kernel<NH>, dim3(az_stream_grid((a->count + 3) / 4, 256)), dim3(256), 0, s, a->x_s,
                     a->pred, a->x_t, a->mean, hp, a->coef, a->count);
}

__global__ __launch_bounds__(256) void silu_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = az_silu(x[i]);
}

__global__ __launch_bounds__(256) void cfg_combine_kernel(float* __restrict__ y, const float* __restrict__ pos,
                                                          const float* __restrict__ neg, const float* __restrict__ g,
                                                          int64_t n) {
  const float k = *g;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = az_add(pos[i], az_mul(k, az_sub(pos[i], neg[i])));
}

__global__ void gather_rows_kernel(float* __restrict__ dst, const float* __restrict__ table,
                                   const int64_t* __restrict__ idx, int64_t nrows, int64_t ncols,
                                   int64_t table_rows) {
  const int64_t r = blockIdx.x;
  int64_t src = idx[r];
  if (src < 0) src = 0;
  if (src >= table_rows) src = table_rows - 1;
  for (int64_t c = threadIdx.x; c < ncols; c += blockDim.x) dst[r * ncols + c] = table[src * ncols + c];
}

__global__ void gather_step_row_kernel(float* __restrict__ dst, const float* __restrict__ table,
                                       const AzStepCoef* coef, int which, int64_t ncols, int64_t table_rows) {
  int64_t src = which == 0 ? coef->time_index : coef->step;
  if (src < 0) src = 0;
  if (src >= table_rows) src = table_rows - 1;
  for (int64_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x)
    dst[c] = table[src * ncols + c];
}

__global__ void coef_c_time_kernel(float* dst, const AzStepCoef* coef) {
  if (threadIdx.x == 0) dst[0] = coef->c_time;
}

// ---- fp64 latents.  Sampler(dtype=torch.float64) makes the reference's schedule scalars fp64 tensors of shape
// (1, ..., 1); multiplied into the latents they promote them to fp64 (azula/denoise.py:306-322, azula/sample.py:210-214,
// 257-259), so the whole elementwise path of such a sampler is fp64 while the backbone keeps its own dtype.  These
// kernels are that path: separately rounded fp64 operations in the reference's association order.
__device__ __forceinline__ double az_dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double az_dadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double az_dsub(double a, double b) { return __dsub_rn(a, b); }

struct AzStepCoef64 {  // the fields of AzStepCoef that the flat transition reads, as doubles (same order)
  double c_in, c_skip, c_out, c_time, alpha_t, alpha_s, k_x, k_eps, c_in_next, clip_lo, clip_hi, guidance;
};

template <bool EPS, bool MEAN>
__global__ __launch_bounds__(256) void transition_f64_kernel(const double* x_t, const double* __restrict__ F,
                                                             const double* __restrict__ eps, double* x_s,
                                                             double* __restrict__ mean_out, int64_t n,
                                                             const AzStepCoef64* coef) {
  const double c_skip = coef->c_skip, c_out = coef->c_out, a_t = coef->alpha_t, a_s = coef->alpha_s, k_x = coef->k_x,
               k_eps = coef->k_eps, lo = coef->clip_lo, hi = coef->clip_hi;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double x = x_t[i];
    double m = az_dadd(az_dmul(c_skip, x), az_dmul(c_out, F[i]));
    const double c = fmin(fmax(m, lo), hi);
    m = m != m ? m : c;
    double xs = az_dmul(a_s, m);
    xs = az_dadd(xs, az_dmul(k_x, az_dsub(x, az_dmul(a_t, m))));
    if (EPS) xs = az_dadd(xs, az_dmul(k_eps, eps[i]));
    x_s[i] = xs;
    if (MEAN) mean_out[i] = m;
  }
}

template <bool Z32>
__global__ __launch_bounds__(256) void axpby_f64_kernel(double* __restrict__ y, const double* __restrict__ a,
                                                        const double* __restrict__ x, const double* __restrict__ b,
                                                        const void* __restrict__ z, int64_t rows, int64_t inner,
                                                        int a_stride) {
  const int64_t n = rows * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = a_stride ? i / inner : 0;
    const double zv = Z32 ? (double)reinterpret_cast<const float*>(z)[i] : reinterpret_cast<const double*>(z)[i];
    y[i] = az_dadd(az_dmul(a[r * a_stride], x[i]), az_dmul(b[r * a_stride], zv));
  }
}

__global__ __launch_bounds__(256) void scale_f64_to_f32_kernel(float* __restrict__ y, const double* __restrict__ x,
                                                               const double* __restrict__ s, int64_t rows,
                                                               int64_t inner, int s_stride) {
  const int64_t n = rows * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (float)az_dmul(s[s_stride ? i / inner : 0], x[i]);  // (c_in * x_t).to(float32), azula/denoise.py:317
}

template <bool CFG, bool EPS>
int launch_flat(const AzTransitionArgs* a, int64_t n, hipStream_t st) {
  const bool per_image = a->f_channels != a->channels;
  const int64_t xstep = per_image ? a->channels * a->inner : 0, fstep = per_image ? a->f_channels * a->inner : 0;
  const unsigned gy = per_image ? (unsigned)a->batch : 1u;
  if (per_image) n = xstep;
  const int64_t n4 = n / 4;
  // one workgroup per TF_UN * 256 float4 (16 KB of every stream), at most 16384 of them (measured: 5.6-5.75 TB/s from
  // 512 workgroups up; the fewest loop iterations win by a little)
  int64_t g64 = (n4 + TF_UN * 256 - 1) / (TF_UN * 256);
  const int64_t gcap = 16384 / gy < 1 ? 1 : 16384 / gy;
  const int grid = (int)(g64 < 1 ? 1 : (g64 > gcap ? gcap : g64));
  const bool xin = a->xin_next != nullptr, mean = a->mean_out != nullptr;
#define AZ_FLAT(X, M)                                                                                          \
  hipLaunchKernelGGL((transition_flat_kernel<CFG, EPS, X, M>), dim3(grid, gy), dim3(256), 0, st, a->x_t, a->F, a->F_neg, \
                     a->eps, a->x_s, a->xin_next, a->mean_out, n4, n, a->coef, xstep, fstep)
  if (xin && mean) AZ_FLAT(true, true);
  else if (xin) AZ_FLAT(true, false);
  else if (mean) AZ_FLAT(false, true);
  else AZ_FLAT(false, false);
#undef AZ_FLAT
  return az_launch_status();
}

template <bool CFG, bool EPS>
int launch_image(const AzTransitionArgs* a, hipStream_t st) {
  const int64_t qps = a->inner / 4;
  if (!a->f_nhwc && a->channels <= 4 && a->nhwc_pad == 4 && a->xin_next != nullptr) {
    const int64_t total = a->batch * qps;
    int64_t g = (total + 256 * TI_UN - 1) / (256 * TI_UN);
    if (g > 16384) g = 16384;
    if (a->mean_out != nullptr)
      hipLaunchKernelGGL((transition_image4_kernel<CFG, EPS, true>), dim3((unsigned)g), dim3(256), 0, st, *a, qps, total);
    else
      hipLaunchKernelGGL((transition_image4_kernel<CFG, EPS, false>), dim3((unsigned)g), dim3(256), 0, st, *a, qps, total);
    return az_launch_status();
  }
  const int grid = az_stream_grid(a->batch * qps, 256);
  if (a->mean_out != nullptr)
    hipLaunchKernelGGL((transition_image_kernel<CFG, EPS, true>), dim3(grid), dim3(256), 0, st, *a, qps);
  else
    hipLaunchKernelGGL((transition_image_kernel<CFG, EPS, false>), dim3(grid), dim3(256), 0, st, *a, qps);
  return az_launch_status();
}

}  // namespace

extern "C" {

int az_version(void) { return AZ_VERSION; }

const char* az_error_string(int code) {
  switch (code) {
    case AZ_OK: return "ok";
    case AZ_E_NULL: return "azula_amd: required pointer is NULL";
    case AZ_E_SHAPE: return "azula_amd: inconsistent or unsupported shape";
    case AZ_E_ALIGN: return "azula_amd: pointer/stride not 16-byte aligned";
    case AZ_E_UNSUPPORTED: return "azula_amd: unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "azula_amd: unknown error";
  }
}

int az_step_begin(AzStepCoef* cur, const AzStepCoef* table, int32_t* step_counter, int32_t n_steps,
                  az_stream_t stream) {
  AZ_REQUIRE(cur && table && step_counter, AZ_E_NULL);
  AZ_REQUIRE(n_steps > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(64), 0, az_s(stream), cur, table, step_counter, n_steps);
  return az_launch_status();
}

int az_step_row_f64(double* cur, const double* table, const int32_t* step_counter, int32_t n_rows, int32_t words,
                    az_stream_t stream) {
  AZ_REQUIRE(cur && table && step_counter, AZ_E_NULL);
  AZ_REQUIRE(n_rows > 0 && words > 0 && words <= 64, AZ_E_SHAPE);
  hipLaunchKernelGGL(step_row_f64_kernel, dim3(1), dim3(64), 0, az_s(stream), cur, table, step_counter, n_rows, words);
  return az_launch_status();
}

int az_transition_f32(const AzTransitionArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->x_t && a->F && a->x_s && a->coef, AZ_E_NULL);
  AZ_REQUIRE(a->batch > 0 && a->channels > 0 && a->inner > 0 && a->f_channels >= a->channels, AZ_E_SHAPE);
  const int64_t n = a->batch * a->channels * a->inner;
  const bool cfg = a->F_neg != nullptr, eps = a->eps != nullptr;
  // flat: every tensor planar in x's layout; F may carry more channels per image if the images stay 16-byte aligned
  const bool flat = !a->f_nhwc && a->nhwc_pad == 0 &&
                    (a->f_channels == a->channels || ((a->channels * a->inner) % 4 == 0 && (a->f_channels * a->inner) % 4 == 0 && a->batch < 65536));
  AZ_REQUIRE(AZ_ALIGNED16(a->x_t) && AZ_ALIGNED16(a->F) && AZ_ALIGNED16(a->x_s), AZ_E_ALIGN);
  AZ_REQUIRE(AZ_ALIGNED16(a->F_neg) && AZ_ALIGNED16(a->eps) && AZ_ALIGNED16(a->xin_next) && AZ_ALIGNED16(a->mean_out),
             AZ_E_ALIGN);
  hipStream_t st = az_s(stream);
  if (flat) {
    if (cfg) return eps ? launch_flat<true, true>(a, n, st) : launch_flat<true, false>(a, n, st);
    return eps ? launch_flat<false, true>(a, n, st) : launch_flat<false, false>(a, n, st);
  }
  AZ_REQUIRE(a->inner % 4 == 0, AZ_E_SHAPE);
  if (a->f_nhwc) AZ_REQUIRE(a->f_channels % 4 == 0, AZ_E_SHAPE);
  if (a->nhwc_pad > 0) AZ_REQUIRE(a->nhwc_pad % 4 == 0 && a->nhwc_pad >= a->channels, AZ_E_SHAPE);
  if (cfg) return eps ? launch_image<true, true>(a, st) : launch_image<true, false>(a, st);
  return eps ? launch_image<false, true>(a, st) : launch_image<false, false>(a, st);
}

int az_scale_f32(float* y, const float* x, const float* s_dev, int64_t n, az_stream_t stream) {
  AZ_REQUIRE(y && x && s_dev, AZ_E_NULL);
  AZ_REQUIRE(n > 0, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(y) && AZ_ALIGNED16(x), AZ_E_ALIGN);
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(scale_kernel, dim3(az_stream_grid(n4 > 0 ? n4 : 1, 256)), dim3(256), 0, az_s(stream), y, x,
                     s_dev, n4, n);
  return az_launch_status();
}

int az_axpby_f32(float* y, const float* a_dev, const float* x, const float* b_dev, const float* z, int64_t rows,
                 int64_t inner, int32_t a_stride, az_stream_t stream) {
  AZ_REQUIRE(y && a_dev && x && b_dev && z, AZ_E_NULL);
  AZ_REQUIRE(rows > 0 && inner > 0 && (a_stride == 0 || a_stride == 1), AZ_E_SHAPE);
  hipLaunchKernelGGL(axpby_kernel, dim3(az_stream_grid(rows * inner, 256)), dim3(256), 0, az_s(stream), y, a_dev, x,
                     b_dev, z, rows, inner, a_stride);
  return az_launch_status();
}

int az_multistep_f32(const AzMultistepArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->x_s && a->pred && a->x_t && a->mean && a->coef, AZ_E_NULL);
  AZ_REQUIRE(a->count > 0 && a->n_hist >= 0 && a->n_hist <= AZ_MULTISTEP_MAX_HIST, AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(a->x_s) && AZ_ALIGNED16(a->pred) && AZ_ALIGNED16(a->x_t) && AZ_ALIGNED16(a->mean),
             AZ_E_ALIGN);
  for (int j = 0; j < a->n_hist; ++j) {
    AZ_REQUIRE(a->hist[j], AZ_E_NULL);
    AZ_REQUIRE(AZ_ALIGNED16(a->hist[j]), AZ_E_ALIGN);
    AZ_REQUIRE(a->hist[j] != a->pred, AZ_E_SHAPE);  // the newest slot must not alias live history
  }
  hipStream_t s = az_s(stream);
  switch (a->n_hist) {
    case 0: launch_multistep<0>(a, s); break;
    case 1: launch_multistep<1>(a, s); break;
    case 2: launch_multistep<2>(a, s); break;
    case 3: launch_multistep<3>(a, s); break;
    case 4: launch_multistep<4>(a, s); break;
    case 5: launch_multistep<5>(a, s); break;
    case 6: launch_multistep<6>(a, s); break;
    default: launch_multistep<7>(a, s); break;
  }
  return az_launch_status();
}

int az_transition_f64(const AzTransitionArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->x_t && a->F && a->x_s && a->coef, AZ_E_NULL);
  AZ_REQUIRE(a->batch > 0 && a->channels > 0 && a->inner > 0, AZ_E_SHAPE);
  // flat form only: every tensor (B, C, inner) fp64 in the same layout; no CFG input, no pre-scaled second output
  AZ_REQUIRE(!a->F_neg && !a->xin_next && !a->f_nhwc && a->nhwc_pad == 0 && a->f_channels == a->channels, AZ_E_UNSUPPORTED);
  const int64_t n = a->batch * a->channels * a->inner;
  const double *x = reinterpret_cast<const double*>(a->x_t), *F = reinterpret_cast<const double*>(a->F),
               *e = reinterpret_cast<const double*>(a->eps);
  double *xs = reinterpret_cast<double*>(a->x_s), *mo = reinterpret_cast<double*>(a->mean_out);
  const AzStepCoef64* cf = reinterpret_cast<const AzStepCoef64*>(a->coef);
  const dim3 grid(az_stream_grid(n, 256));
  hipStream_t st = az_s(stream);
  if (e && mo) hipLaunchKernelGGL((transition_f64_kernel<true, true>), grid, dim3(256), 0, st, x, F, e, xs, mo, n, cf);
  else if (e) hipLaunchKernelGGL((transition_f64_kernel<true, false>), grid, dim3(256), 0, st, x, F, e, xs, mo, n, cf);
  else if (mo) hipLaunchKernelGGL((transition_f64_kernel<false, true>), grid, dim3(256), 0, st, x, F, e, xs, mo, n, cf);
  else hipLaunchKernelGGL((transition_f64_kernel<false, false>), grid, dim3(256), 0, st, x, F, e, xs, mo, n, cf);
  return az_launch_status();
}

int az_axpby_f64(double* y, const double* a_dev, const double* x, const double* b_dev, const void* z, int32_t z_is_f32,
                 int64_t rows, int64_t inner, int32_t a_stride, az_stream_t stream) {
  AZ_REQUIRE(y && a_dev && x && b_dev && z, AZ_E_NULL);
  AZ_REQUIRE(rows > 0 && inner > 0 && (a_stride == 0 || a_stride == 1), AZ_E_SHAPE);
  const dim3 grid(az_stream_grid(rows * inner, 256));
  if (z_is_f32)
    hipLaunchKernelGGL(axpby_f64_kernel<true>, grid, dim3(256), 0, az_s(stream), y, a_dev, x, b_dev, z, rows, inner, a_stride);
  else
    hipLaunchKernelGGL(axpby_f64_kernel<false>, grid, dim3(256), 0, az_s(stream), y, a_dev, x, b_dev, z, rows, inner, a_stride);
  return az_launch_status();
}

int az_scale_f64_to_f32(float* y, const double* x, const double* s_dev, int64_t rows, int64_t inner, int32_t s_stride,
                        az_stream_t stream) {
  AZ_REQUIRE(y && x && s_dev, AZ_E_NULL);
  AZ_REQUIRE(rows > 0 && inner > 0 && (s_stride == 0 || s_stride == 1), AZ_E_SHAPE);
  hipLaunchKernelGGL(scale_f64_to_f32_kernel, dim3(az_stream_grid(rows * inner, 256)), dim3(256), 0, az_s(stream), y, x,
                     s_dev, rows, inner, s_stride);
  return az_launch_status();
}

int az_silu_f32(float* y, const float* x, int64_t n, az_stream_t stream) {
  AZ_REQUIRE(y && x, AZ_E_NULL);
  AZ_REQUIRE(n > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(silu_kernel, dim3(az_stream_grid(n, 256)), dim3(256), 0, az_s(stream), y, x, n);
  return az_launch_status();
}

int az_cfg_combine_f32(float* y, const float* pos, const float* neg, const float* g_dev, int64_t n,
                       az_stream_t stream) {
  AZ_REQUIRE(y && pos && neg && g_dev, AZ_E_NULL);
  AZ_REQUIRE(n > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(az_stream_grid(n, 256)), dim3(256), 0, az_s(stream), y, pos, neg, g_dev,
                     n);
  return az_launch_status();
}

int az_gather_rows_f32(float* dst, const float* table, const int64_t* idx, int64_t nrows, int64_t ncols,
                       int64_t table_rows, az_stream_t stream) {
  AZ_REQUIRE(dst && table && idx, AZ_E_NULL);
  AZ_REQUIRE(nrows > 0 && ncols > 0 && table_rows > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nrows), dim3(256), 0, az_s(stream), dst, table, idx, nrows,
                     ncols, table_rows);
  return az_launch_status();
}

int az_gather_step_row_f32(float* dst, const float* table, const AzStepCoef* coef, int32_t which, int64_t ncols,
                           int64_t table_rows, az_stream_t stream) {
  AZ_REQUIRE(dst && table && coef, AZ_E_NULL);
  AZ_REQUIRE(ncols > 0 && table_rows > 0, AZ_E_SHAPE);
  hipLaunchKernelGGL(gather_step_row_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, az_s(stream), dst,
                     table, coef, which, ncols, table_rows);
  return az_launch_status();
}

int az_coef_c_time_f32(float* dst, const AzStepCoef* coef, az_stream_t stream) {
  AZ_REQUIRE(dst && coef, AZ_E_NULL);
  hipLaunchKernelGGL(coef_c_time_kernel, dim3(1), dim3(64), 0, az_s(stream), dst, coef);
  return az_launch_status();
}

}  // extern "C"
