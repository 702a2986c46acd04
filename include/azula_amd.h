/*
 * azula_amd -- C ABI of the MI355X (gfx950) sampling hot path.
 *
 * The reference (probabilists/azula, /root/reference) is pure Python and has NO FFI: its
 * "operator boundary" is the duck-typed protocol Schedule / Denoiser / Sampler
 * (azula/noise.py:49-63, azula/denoise.py:97-114, azula/sample.py:54-161).  This header is
 * the boundary a maintainer would bind (ctypes stub in INTEGRATION.md) to replace the ATen
 * op sequences that protocol dispatches per sampling step.  Each entry point cites the
 * reference lines whose arithmetic it replaces.
 *
 * Contract of every function below:
 *   - plain C, raw DEVICE pointers + sizes + POD structs + a hipStream_t (passed as void*);
 *   - returns 0 on success, a negative AZ_E_* argument error, or a positive hipError_t;
 *   - never allocates, never synchronises, never throws, keeps no mutable global state,
 *     is re-entrant and hipGraph-capturable (workspace is supplied by the caller);
 *   - tensors are fp32 unless the entry point's name says otherwise (`_f64`: fp64 latents of a Sampler(dtype=float64);
 *     `_bf16_f32` / `_f16_f32`: half-precision MFMA operands packed by az_pack_conv_weight_half_f32, fp32 activations and
 *     accumulation).  "NHWC" tensors have a channel stride that is a multiple of 4 floats (zero-filled pad channels), so
 *     every access is a 16-byte vector.
 */
#ifndef AZULA_AMD_H
#define AZULA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_VERSION 1

enum {
  AZ_OK = 0,
  AZ_E_NULL = -1,      /* required pointer is NULL            */
  AZ_E_SHAPE = -2,     /* inconsistent / unsupported shape    */
  AZ_E_ALIGN = -3,     /* pointer or stride not 16-B aligned  */
  AZ_E_UNSUPPORTED = -4
};

typedef void* az_stream_t; /* hipStream_t */

int az_version(void);
const char* az_error_string(int code);

/* ------------------------------------------------------------------ per-step scalars
 * One row per sampling step, computed on the HOST with torch-CPU 0-d ops in the
 * reference's op order (azula/noise.py:125-129, azula/denoise.py:309-312,
 * azula/plugins/adm/__init__.py:109-114, azula/sample.py:249-253) and uploaded once --
 * the analogue of the reference's single `time_pairs` H2D copy (azula/sample.py:151).   */
typedef struct AzStepCoef {
  float c_in;      /* backbone input scale for THIS step                                  */
  float c_skip;    /* mean = c_skip * x_t + c_out * F                                      */
  float c_out;
  float c_time;    /* Karras: log(sigma_t / alpha_t)                                       */
  float alpha_t;
  float alpha_s;
  float k_x;       /* sigma_s * sqrt(1 - tau) / sigma_t                                    */
  float k_eps;     /* sigma_s * sqrt(tau)                                                  */
  float c_in_next; /* c_in of the NEXT step (pre-scaled second output), 0 on the last step */
  float clip_lo;   /* mean clip (ADM eval, plugins/adm/__init__.py:133-134); -inf/+inf = off */
  float clip_hi;
  float guidance;  /* CFG strength (azula/guidance/cfg.py:63-65)                           */
  int32_t time_index; /* ADM: searchsorted(sigmas, sigma_t * c_in)                          */
  int32_t step;       /* step number (informational)                                        */
  float pad[2];
} AzStepCoef; /* 64 bytes */

/* cur[0] = table[*step_counter]; *step_counter += 1.  One thread; first node of a step
 * graph, so that the same captured graph can be replayed for every step.               */
int az_step_begin(AzStepCoef* cur, const AzStepCoef* table, int32_t* step_counter, int32_t n_steps,
                  az_stream_t stream);

/* ------------------------------------------------------------------ K1: fused transition
 * Replaces, in ONE pass over the latent (12 B/element for DDIM eta=0, 16 B with eps):
 *   azula/denoise.py:322            mean = c_skip*x_t + c_out*F   [ADM: clip, plugins/adm/__init__.py:126-134]
 *   azula/guidance/cfg.py:63-65     mean = mean_p + g*(mean_p - mean_n)        (if F_neg != NULL)
 *   azula/sample.py:257-259         x_s = alpha_s*mean + k_x*(x_t - alpha_t*mean) + k_eps*eps
 *   azula/denoise.py:317 (next step) xin = c_in_next * x_s                      (if xin_next != NULL)
 * Arithmetic is done with separately rounded mul/add in the reference's association order,
 * so the result is bit-identical to the torch-CPU op sequence for the same inputs.
 *
 * Flat form: x_t, F, F_neg, eps, x_s, xin_next all have n elements.
 * F may have `f_channels` >= channels per sample (ADM learn_var: 6 of which 3 are used):
 *   F element (b, c, i) is at F[(b*f_channels + c)*inner + i], x at (b*channels + c)*inner + i. */
typedef struct AzTransitionArgs {
  const float* x_t;
  const float* F;      /* backbone output (positive branch)            */
  const float* F_neg;  /* backbone output (negative CFG branch) or NULL */
  const float* eps;    /* noise or NULL (treated as 0: DDIM eta = 0)    */
  float* x_s;          /* may alias x_t                                 */
  float* xin_next;     /* optional: c_in_next * x_s, same layout as x_s unless nhwc_pad > 0 */
  float* mean_out;     /* optional: posterior mean (for Denoiser.forward), same layout as x_s */
  int64_t batch, channels, inner; /* x is (batch, channels, inner)      */
  int64_t f_channels;  /* channels of F per sample (>= channels)        */
  int32_t f_nhwc;      /* 1: F is NHWC with channel stride f_channels (backbone-native layout) */
  int32_t nhwc_pad;    /* >0: xin_next is written NHWC with this channel stride (zero pad)    */
  const AzStepCoef* coef; /* DEVICE pointer                              */
} AzTransitionArgs;
int az_transition_f32(const AzTransitionArgs* args, az_stream_t stream);

/* ---- linear multistep update ------------------------------------------------
 * Replaces the per-step tensor arithmetic of the Adams-Bashforth sampler family:
 *   zABSampler.__call__   azula/sample.py:519-546     vABSampler.__call__   azula/sample.py:593-620
 *   zEABSampler.__call__  azula/sample.py:688-715     xEABSampler.__call__  azula/sample.py:797-821
 *   REABSampler.__call__  azula/sample.py:915-950
 * which all have the form (prediction linear in x_t and the posterior mean, history of <= order
 * predictions, x_s linear in x_t and the history):
 *   pred = a x_t + b mean                      -> written to `pred` (its history slot)
 *   x_s  = p x_t + sum_j w_j hist[j] + w_new pred
 * `coef` is a DEVICE pointer to 4 + n_hist floats [a, b, p, w_new, w_0 .. w_{n_hist-1}] (the
 * fp64 Vandermonde solves of `_adams_bashforth` stay on the host, azula/sample.py:486-508).
 * x_s may alias x_t; pred must not alias any hist[j].  All pointers 16-byte aligned.          */
#define AZ_MULTISTEP_MAX_HIST 7
typedef struct AzMultistepArgs {
  float* x_s;
  float* pred;
  const float* x_t;
  const float* mean;
  const float* hist[AZ_MULTISTEP_MAX_HIST]; /* older predictions, oldest first */
  const float* coef;                        /* DEVICE pointer                  */
  int64_t count;                            /* elements                        */
  int32_t n_hist;                           /* 0 .. AZ_MULTISTEP_MAX_HIST      */
  int32_t pad_;
} AzMultistepArgs;
int az_multistep_f32(const AzMultistepArgs* args, az_stream_t stream);

/* y = s * x with s read from device memory (azula/denoise.py:317 c_in * x_t, generic backbones). */
int az_scale_f32(float* y, const float* x, const float* s_dev, int64_t n, az_stream_t stream);

/* y = silu(x) = x / (1 + exp(-x)): the activation in front of every adaLN projection of the conditioning
 * vector (plugins/jit/_src/model.py:175-177,199-201), applied once so that all projections become one GEMM. */
int az_silu_f32(float* y, const float* x, int64_t n, az_stream_t stream);

/* y[r, i] = a[r*a_stride] * x[r, i] + b[r*a_stride] * z[r, i]  with a, b in device memory
 * (a_stride = 0: one scalar pair for all rows; 1: one pair per row = per-sample times t of
 * shape (B,), azula/denoise.py:306-307).  Separately rounded mul/mul/add as
 * azula/denoise.py:322  mean = c_skip * x_t + c_out * output.                               */
int az_axpby_f32(float* y, const float* a_dev, const float* x, const float* b_dev, const float* z, int64_t rows,
                 int64_t inner, int32_t a_stride, az_stream_t stream);

/* y = pos + g * (pos - neg), g in device memory: classifier-free guidance outside a fused
 * sampler (azula/guidance/cfg.py:63-65), same association and rounding as the reference.   */
int az_cfg_combine_f32(float* y, const float* pos, const float* neg, const float* g_dev, int64_t n,
                       az_stream_t stream);

/* ------------------------------------------------------------------ layout
 * NCHW (B,C,H,W) -> NHWC with channel stride cs (>= C, multiple of 4), scaled by *scale_dev
 * (NULL = 1).  Pad channels are written as 0.  And back.                                  */
int az_nchw_to_nhwc_f32(float* dst, const float* src, const float* scale_dev, int64_t B, int64_t C, int64_t HW,
                        int64_t cs, az_stream_t stream);
/* Nearest upsampling by an arbitrary integer factor per axis, cropped to (Hout, Wout): dst[b, y, x, :] = src[b, y / sh, x / sw, :]
 * on NHWC activations (channel stride cs).  Replaces torch.nn.Upsample(scale_factor=stride, mode="nearest") + torch.narrow of
 * azula/nn/unet.py:186,250-252 for strides that are not powers of two (those are folded into the merge convolution). */
int az_upsample_nearest_f32(float* dst, const float* src, int64_t B, int64_t Hin, int64_t Win, int64_t cs, int32_t sh,
                            int32_t sw, int64_t Hout, int64_t Wout, az_stream_t stream);

int az_nhwc_to_nchw_f32(float* dst, const float* src, int64_t B, int64_t C, int64_t HW, int64_t cs,
                        az_stream_t stream);

/* ------------------------------------------------------------------ K5 (small M): linear + activation
 * y[m, n] = out_act( sum_k in_act(x[m, k]) * W[n, k] + bias[n] ),  M small (modulation / time MLPs:
 * azula/nn/unet.py:65-70, azula/nn/dit.py:58-63, plugins/adm/_src/unet.py:196-202,460-464).
 * act: 0 none, 1 SiLU.  K must be a multiple of 4 or K == 1.  ldy = row stride of y.        */
int az_linear_small_f32(float* y, int64_t ldy, const float* x, int64_t ldx, const float* W, const float* bias,
                        int64_t M, int64_t N, int64_t K, int32_t in_act, int32_t out_act, az_stream_t stream);
typedef struct AzLinearGroup {
  float* y;          /* (M, ldy) */
  const float* x;    /* (M, ldx); K % 4 == 0, ldx % 4 == 0, 16-byte aligned */
  const float* W;    /* (N, K) row-major, 16-byte aligned */
  const float* bias; /* (N) or NULL */
  int64_t ldy, ldx;
  int32_t N, K;
} AzLinearGroup; /* 56 bytes */
/* The same operation for `ngroups` independent (x, W, bias, y) quadruples sharing M and the activations, in ONE
 * launch: `groups_dev` is a DEVICE array of descriptors, max_n = max N over the groups.  Used to hoist all
 * per-block modulation MLPs (azula/nn/unet.py:65-70, one per UNetBlock) to the front of a forward.      */
int az_linear_small_grouped_f32(const AzLinearGroup* groups_dev, int32_t ngroups, int32_t max_n, int64_t M,
                                int32_t in_act, int32_t out_act, az_stream_t stream);

/* dst[r, :] = table[idx[r], :]  (label_emb / per-step embedding rows).  idx is int64 on device. */
int az_gather_rows_f32(float* dst, const float* table, const int64_t* idx, int64_t nrows, int64_t ncols,
                       int64_t table_rows, az_stream_t stream);
/* dst[0, :] = table[coef->time_index or coef->step, :] (which: 0 = time_index, 1 = step)       */
int az_gather_step_row_f32(float* dst, const float* table, const AzStepCoef* coef, int32_t which, int64_t ncols,
                           int64_t table_rows, az_stream_t stream);
/* dst[0] = coef->c_time (feeds the time-embedding MLP of a Karras-wrapped backbone)            */
int az_coef_c_time_f32(float* dst, const AzStepCoef* coef, az_stream_t stream);

/* ------------------------------------------------------------------ K2: GroupNorm (+ modulation, + SiLU)
 * Three launches over an NHWC tensor x (B, HW, cs):
 *  (1) az_groupnorm_stats_f32: per (b, pixel-chunk, group) partial (count, mean, M2), Chan-combinable;
 *  (2) az_groupnorm_finalize_f32: per (b, c): S = rstd*w*(1+a), T = (bias - mean*rstd*w)*(1+a) + sh
 *      so that  y = x*S + T  ==  (GN(x)*w + bias) * (1 + a) + sh
 *      (azula/nn/unet.py:91 with w=1,bias=0;  plugins/adm/_src/nn.py:87 + _src/unet.py:179-183,239-243);
 *  (3) az_affine_act_f32: y = act(x*S + T), optionally 2x2 average pooled (ADM Downsample,
 *      _src/unet.py:133).
 * partials: float[B * nchunks * groups * 4].                                                   */
/* x1 / c0s: optional second source -- the input is the channel concatenation [x (c0s) | x1 (cs - c0s)]
 * (ADM decoder: GroupNorm over cat([h, skip]), plugins/adm/_src/unet.py:631,179); NULL / 0 otherwise.  */
int az_groupnorm_stats_f32(float* partials, const float* x, const float* x1, int64_t c0s, int64_t B, int64_t HW,
                           int64_t C, int64_t cs, int32_t groups, int32_t nchunks, az_stream_t stream);
typedef struct AzNormFinalizeArgs {
  float* S;               /* (B, cs) */
  float* T;               /* (B, cs) */
  const float* partials;  /* from az_groupnorm_stats_f32 */
  const float* weight;    /* (C) or NULL (= 1) */
  const float* bias;      /* (C) or NULL (= 0) */
  const float* scale;     /* a: (C) if scale_bstride == 0 else (B, scale_bstride...) ; NULL = 0 */
  const float* shift;     /* same addressing as scale */
  int64_t scale_bstride;  /* elements between consecutive batch rows of scale/shift (0 = shared) */
  int64_t B, C, cs;
  int32_t groups, nchunks;
  float eps;
  /* partials produced by convolution epilogues (AzConvArgs.gn_quads) instead of az_groupnorm_stats_f32:
   * quads_per_group > 0 selects the layout (B, nchunks, quads, 4) with `quads0` channel quads in `partials` and, for a
   * two-source input (channel concatenation), the rest in `partials1`; group g folds quads [g qpg, (g + 1) qpg). */
  const float* partials1;
  int32_t quads_per_group, quads0;
  int32_t nchunks1;       /* partials per image in `partials1` (0 = nchunks): the two producers may chunk differently */
  int32_t reserved0;
} AzNormFinalizeArgs;
int az_groupnorm_finalize_f32(const AzNormFinalizeArgs* args, az_stream_t stream);
/* pool: 0 none, 1 = 2x2 average pool of act(.) (needs H, W even; dst is (B, H/2*W/2, cs)), 2 = 1x2 (width only: AvgPool1d(2) of
 * a signal held as a one-row image, plugins/adm/_src/nn.py:64-77 with dims = 1; dst is (B, H*W/2, cs)). */
int az_affine_act_f32(float* y, const float* x, const float* x1, int64_t c0s, const float* S, const float* T,
                      int64_t B, int64_t H, int64_t W, int64_t cs, int32_t act, int32_t pool, az_stream_t stream);

/* Row norms over the channel axis of an NHWC / token tensor (rows, cs), C real channels:
 *   kind 0: azula layer_norm, UNBIASED variance, no affine (azula/nn/layers.py:152-155);
 *   kind 1: RMS norm (azula/nn/layers.py:193-195, torch.nn.RMSNorm in azula/nn/dit.py:52-55);
 * optionally times a learned per-channel `weight` (NULL = none; the RMSNorm of the JiT plugin,
 * plugins/jit/_src/util.py:149-163), followed by y = n * (1 + a[b, c]) + sh[b, c]
 * (azula/nn/unet.py:91, azula/nn/dit.py:107, plugins/jit/_src/model.py:12-13).
 * rows_per_batch maps a row to its batch index for the modulation lookup.                      */
int az_rownorm_mod_f32(float* y, const float* x, const float* weight, const float* scale, const float* shift,
                       int64_t scale_bstride,
                       int64_t rows, int64_t rows_per_batch, int64_t C, int64_t cs, int32_t kind, float eps,
                       az_stream_t stream);
/* az_groupnorm_stats_f32 and az_affine_act_f32 on tensors held in a 2-byte type -- dtype 1: bfloat16, 2: IEEE half; x, x1 and y
 * alike: the activations of a module cast to half precision (azula/denoise.py:314-320; torch.nn.GroupNorm on a half tensor
 * computes its statistics in fp32 as well).  Strides in elements, multiples of 8; y rounded to nearest even.                  */
int az_groupnorm_stats_h16(float* partials, const void* x, const void* x1, int64_t c0s, int64_t B, int64_t HW, int64_t C,
                           int64_t cs, int32_t groups, int32_t nchunks, int32_t dtype, az_stream_t stream);
int az_affine_act_h16(void* y, const void* x, const void* x1, int64_t c0s, const float* S, const float* T, int64_t B, int64_t H,
                      int64_t W, int64_t cs, int32_t act, int32_t pool, int32_t dtype, az_stream_t stream);
/* The same on rows held in a 2-byte type -- dtype 1: bfloat16, 2: IEEE half; x and y alike: the activations of a module cast to
 * half precision (azula/denoise.py:314-320; torch's RMSNorm / the reference's layer_norm on a half tensor).  Statistics, gain and
 * modulation in fp32, output rounded to nearest even.  C % 8 == 0, C == cs, C <= 4096 (else AZ_E_UNSUPPORTED).                 */
int az_rownorm_mod_h16(void* y, const void* x, const float* weight, const float* scale, const float* shift,
                       int64_t scale_bstride, int64_t rows, int64_t rows_per_batch, int64_t C, int64_t cs, int32_t kind,
                       float eps, int32_t dtype, az_stream_t stream);

/* ------------------------------------------------------------------ K3/K5: implicit-GEMM convolution
 * out[b, oh, ow, co] = epilogue( bias[co] + sum_{ky,kx,ci} in[b, oh*s+ky-p, ow*s+kx-p, ci] * W[co, ci, ky, kx] )
 * on gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain).  Replaces
 * F.conv2d 3x3 / 1x1 (azula/nn/unet.py:79-83,175-200; plugins/adm/_src/unet.py:182,207,213,215,471,602),
 * nn.Linear on token tensors (ksize = 1: azula/nn/dit.py:88-93,169-170; azula/nn/attention.py:45-46)
 * and Conv1d k=1 (plugins/adm/_src/unet.py:277,285).
 * The input is the channel concatenation [src0 (c0) | src1 (c1)] (skip concat: azula/nn/unet.py:257,
 * plugins/adm/_src/unet.py:631); either source may be read through nearest x2 upsampling
 * (azula/nn/unet.py:187, plugins/adm/_src/unet.py:104-106) without materialising it.
 * Weights are pre-packed once as [ky*ks+kx][cout_s][cin_s] (az_pack_conv_weight_f32).
 * Epilogue: v = acc + bias; v = act(v); if (gate) v = gate[b*gate_bstride + co] * v; if (res) v += res;
 *           (azula/nn/unet.py:93 x + c*y; plugins/adm/_src/unet.py:247,296 skip + h).           */
typedef struct AzConvArgs {
  const float* src0; /* NHWC (B, h0, w0, c0s) */
  const float* src1; /* NHWC (B, h1, w1, c1s) or NULL */
  int32_t c0s, c1s;  /* channel strides (multiples of 4); input channels = c0s + c1s */
  int32_t up0, up1;  /* log2 of the nearest upsampling the source is read through (0: none, 1: x2, 2: x4, ... <= 4) */
  int32_t h0, w0, h1, w1; /* stored spatial dims of each source */
  int32_t batch, hin, win; /* logical input dims (after upsampling / narrowing) */
  const float* weight;     /* packed [ks*ks][cout_s][cin_s], cin_s = c0s + c1s */
  const float* bias;       /* (cout_s) or NULL */
  int32_t cout_s;          /* output channel stride (multiple of 4) */
  int32_t ksize, stride, pad;
  int32_t hout, wout;
  int32_t act;             /* 0 none, 1 SiLU, 2 ReLU, 3 ReLU^2, 4 SwiGLU over interleaved pairs: dst gets cout_s / 2 channels per
                              pixel, y[c] = x[2c] * silu(x[2c+1]) (no gate / res / dst_nchw; cout_s % 8 == 0), 5 q / k preparation
                              of a fused qkv projection (the qk_* fields at the end of this struct; az_conv2d_f32 family only),
                              6 SiLU of the sum with the residual: y = silu(conv + bias + res) (res required; no gate /
                              dst_nchw / gn_quads) -- the last depth tap of a Conv3d -> SiLU pair */
  const float* gate;       /* optional (…, cout_s) */
  int64_t gate_bstride;    /* 0 = shared across the batch */
  const float* res;        /* optional residual, NHWC (B, hres, wres, cout_s) */
  int32_t res_up;          /* 1: residual is read through nearest x2 upsampling (ADM up block) */
  int32_t hres, wres;
  int32_t res_bcast;       /* 1: residual is (hout, wout, cout_s), shared by the batch (positional embedding) */
  float* dst;              /* NHWC (B, hout, wout, cout_s), or NCHW (B, dst_c, hout, wout) */
  int32_t dst_nchw;        /* 1: write NCHW with dst_c real channels */
  int32_t dst_c;
  int32_t splitk;          /* >= 1; > 1 needs workspace of splitk * B*hout*wout * cout_s floats */
  float* workspace;
  int32_t pad_mode;        /* 0: zero padding; 1: circular ("periodic", azula/nn/unet.py:175-180): taps wrap around the map */
  int32_t gn_chunks;       /* with gn_quads: partials per image -- splitk 1 (az_conv2d_winograd_f32 only): tile blocks of 64 Winograd
                            * tiles; splitk > 1 (any conv entry point): chunks of ceil(hout * wout / gn_chunks) pixels */
  float* gn_quads;         /* optional: GroupNorm moments of the OUTPUT, (batch, gn_chunks, cout_s / 4, 4) floats = (n, mean, M2, 0)
                            * per image, chunk and channel quad -- from the Winograd kernel's epilogue (splitk 1, cout_s % 64 == 0,
                            * 64 | tiles per image) or from the split-K combine kernel (splitk > 1); consumed by
                            * az_groupnorm_finalize_f32 (quads_per_group), so that the normalisation that follows needs no
                            * statistics pass over the tensor */
  int32_t aniso;           /* 1: the WIDTH axis has its own stride / upsampling factors (azula/nn/unet.py:159-186 with a
                            * stride sequence such as (2, 1)); 0: `stride`, `up0`, `up1` apply to both axes.  Direct kernels
                            * (az_conv2d_f32 / _bf16_f32 / _f16_f32 / _x3_f32) only: the Winograd entries return UNSUPPORTED */
  int32_t stride_w;        /* with aniso: stride along the width (`stride` is then the height's) */
  int32_t up0_w, up1_w;    /* with aniso: log2 upsampling of each source along the width */
  const float* in_affine;  /* optional, az_conv2d_winograd_f32 only: (2, batch, c0s) floats [scale | shift]; the convolution's input
                            * is x * scale[b, c] + shift[b, c] (in_act 1: SiLU of that), evaluated on the raw patch values inside the gather --
                            * the GroupNorm apply pass that would write the normalised tensor (az_affine_act_f32) disappears.
                            * Padding stays ZERO (the reference pads the normalised tensor: azula/nn/unet.py:85-92,
                            * plugins/adm/_src/unet.py:196-203).  Needs a single source, c0s % 8 == 0, up0 == 0 */
  int32_t in_act;          /* with in_affine: 0 none, 1 SiLU */
  int32_t src_dtype;       /* az_conv2d_bf16_f32 / az_conv2d_f16_f32 only (0 elsewhere): 0 = the sources hold fp32 (rounded to the operand type
                            * per tile), 1 = they hold the launch's 2-byte operand type already -- the activations of a module cast to
                            * half precision live in HBM in the module's type (azula/denoise.py:314-320); channel strides are then
                            * multiples of 8, `src0` / `src1` are the tensors' addresses and the strides stay in ELEMENTS */
  int32_t depth;           /* > 0 (az_conv2d_f32, az_conv2d_winograd_f32; every source holds `batch` planes): the `batch` images are the planes of
                            * batch / depth volumes, and this launch is ONE DEPTH TAP of a 3-D convolution (azula/nn/layers.py:25-68
                            * with spatial = 3): image b reads source plane b + depth_shift, taken as zeros where
                            * (b % depth) + depth_shift falls outside [0, depth) -- the zero padding along the depth axis; a gate
                            * must be shared by the batch (gate_bstride 0).  The taps accumulate through `res` = `dst` */
  int32_t depth_shift;
  /* act 5 (az_conv2d_f32 and its half / x3 forms, NHWC destination, no gate / res / gn_quads, splitk 1): the output is a fused
   * q | k | v projection laid out '(3 H C)' (azula/nn/attention.py:90, plugins/jit/_src/model.py) with cout_s = 3 * qk_heads *
   * qk_head_dim, and the epilogue prepares q and k for the attention kernel ONCE -- per (token, head): RMS norm over the head's
   * channels (qk_rmsnorm 1; azula/nn/attention.py:92-93), the learned gains qk_q_weight / qk_k_weight ((head_dim) each or NULL),
   * the rotation of adjacent (re, im) channel pairs by qk_rope_cos / qk_rope_sin ((qk_tokens, qk_heads * head_dim / 2) or NULL;
   * attention.py:94-95) -- instead of every workgroup of az_attention_f32 repeating it for all keys of its head.  The v third
   * only gets the bias.  head_dim 32, 64 or 128; a pixel index is batch * qk_tokens + token. */
  int32_t qk_head_dim;
  int32_t qk_heads;
  int32_t qk_tokens;
  int32_t qk_rmsnorm;
  float qk_eps;
  int32_t qk_reserved;
  const float* qk_q_weight;
  const float* qk_k_weight;
  const float* qk_rope_cos;
  const float* qk_rope_sin;
  int32_t depth_wrap;      /* with depth: 1 = circular padding along the depth axis (azula/nn/layers.py:25-68 with
                            * padding_mode "circular": a tap that leaves the volume reads the plane at its other end) */
  int32_t dst_dtype;       /* az_conv2d_bf16_f32 / az_conv2d_f16_f32 only (0 elsewhere): 1 = `dst` AND `res` hold the launch's 2-byte operand type
                            * (rounded to nearest even on the way out; bias, activation, gate, residual arithmetic stays fp32; GroupNorm
                            * moments are those of the rounded values); the planar destination (dst_nchw) and the split-K workspace are
                            * always fp32 */
  float w_scale;           /* az_conv2d_f16x2_f32 / az_conv2d_winograd_f16x2_f32 only (ignored elsewhere): the power of two the packing
                            * (az_pack_conv_weight_f16x2_f32 / az_winograd_pack_filter_f16x2_f32) multiplied the weights by; the kernels
                            * multiply their accumulators by 1 / (activation scale * w_scale), exactly (the activation scale:
                            * AZ_F16X2_IN_SCALE, or the one derived from in_absmax0 / in_absmax1 below) */
  int32_t reserved1;
  const float* in_absmax0; /* az_conv2d_f16x2_f32 / az_conv2d_winograd_f16x2_f32 only, optional: AZ_ABSMAX_SLOTS floats written by az_absmax_f32 over */
  const float* in_absmax1; /* src0 (and src1): the kernel then scales its activation operand by the power of two that puts the LARGEST
                            * magnitude of the sources (x 4 in the Winograd form, whose V sums four pixels) into [2^13, 2^14) instead of
                            * by the fixed AZ_F16X2_IN_SCALE -- no stated range any more (any finite fp32 input), full relative precision
                            * for everything within 2^-27 of the largest value.  For inputs whose magnitude is not bounded by
                            * construction (residual / input streams).  NULL: the fixed scale.  Non-finite maxima fall back to it. */
} AzConvArgs;
/* Narrow outputs (cout_s == 4, 3x3 stride 1 pad 1, one un-upsampled source with c0s % 16 == 0: the image head of
 * azula/nn/unet.py) run a VALU kernel instead of the 128-cout MFMA tile; splitk is ignored there.              */
int az_conv2d_f32(const AzConvArgs* args, az_stream_t stream);
/* The image stem -- the network's first convolution, 3x3 / stride 1 / pad 1 over <= 4 input channels (azula/nn/unet.py:142-150
 * `UNet.in_conv`-style first layer, plugins/adm/_src/unet.py:469-471 `input_blocks[0]`) -- reading its source PLANAR, i.e. the
 * latent (batch, c0s, hin, win) exactly as azula's samplers hold it, so that no NHWC copy of the latent exists: src0 planar
 * with c0s = 1..4 channels; weight = (3, 3, c0s, cout_s) floats; bias, act 0 / 1, pad_mode, gn_quads (gn_chunks =
 * ceil(hin / 8) * ceil(win / 32)) as in az_conv2d_f32; every other option must be unset. */
int az_conv2d_stem_f32(const AzConvArgs* args, az_stream_t stream);
/* The same operation with bf16 / f16 MFMA operands (v_mfma_f32_32x32x16_{bf16,f16}, 16x the fp32 MFMA rate) and fp32
 * accumulation, for backbones cast to half precision (azula/denoise.py:314-320 casts c_in x_t to the module dtype;
 * the reference's own tolerance for that mode is tests/test_nn_unet.py:78-91).  `weight` = az_pack_conv_weight_half_f32
 * output; src / res / dst stay fp32 tensors -- activations are rounded to the operand type while they are staged.   */
int az_conv2d_bf16_f32(const AzConvArgs* args, az_stream_t stream);
int az_conv2d_f16_f32(const AzConvArgs* args, az_stream_t stream);
/* fp32 operands on the bf16 matrix pipe ("bf16x3", the default of the direct contractions since round 4; AZ_FP32_MFMA=native opts out): each fp32 value is split exactly into
 * three bf16 pieces and a product is the six largest of the nine partial products, accumulated in fp32
 * (6 x v_mfma_f32_32x32x16_bf16 per 16 channels = 0.375 x the time of 8 x v_mfma_f32_32x32x2_f32).  Error vs fp64 at the
 * level of the fp32 kernel and below the Winograd form's.  `weight` = az_pack_conv_weight_x3_f32 output; everything else
 * (fp32 src / res / dst, the fused epilogue) as az_conv2d_f32.  Same reference op: azula/nn/layers.py:48-55 ConvNd.
 * Domain: finite operands (an Inf operand splits into NaN pieces: Inf in -> NaN out, where the fp32 MFMA gives Inf); operands
 * below ~2^-110 keep ~16 significant bits (their low pieces are subnormal bf16 values).                                  */
int az_conv2d_x3_f32(const AzConvArgs* args, az_stream_t stream);
/* The split-K az_conv2d_x3_f32 wants for a filled descriptor (splitk / workspace not read): its 256 x 256-tile kernel runs one
 * workgroup per CU and splits a deep K (the 3072 -> 768 token projections) until its rounds are whole; az_conv2d_suggest_splitk's
 * value wherever the 128 x 128 kernel takes the launch.  No reference counterpart. */
int az_conv2d_x3_suggest_splitk(const AzConvArgs* args);
/* Winograd F(2x2,3x3) form of the same operation for ksize = 3, stride = 1, pad = 1: 2.25x fewer
 * multiplies in exact fp32 (transforms only add/subtract; the input transform, the 16 frequency
 * GEMMs and the output transform + epilogue are ONE kernel).  `weight` must be the host-side
 * filter transform U = G g G^T packed [chunk of 8 cin][cout block of 64][16][64][8]
 * (azula_amd/engine.py: Builder.pack_winograd); all other fields as az_conv2d_f32.             */
int az_conv2d_winograd_f32(const AzConvArgs* args, az_stream_t stream);
int az_conv2d_winograd_suggest_splitk(int64_t batch, int32_t hout, int32_t wout, int32_t cout_s, int32_t cin_s);
/* The same F(2x2,3x3) convolution with its 16 frequency GEMMs on the bf16 matrix pipe at fp32 accuracy (csrc/wino_x3.hip): U (at
 * pack time) and V = B^T d B (in the kernel) are split EXACTLY into three bf16 pieces and a product is the six largest of the
 * nine partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 0.375 x the matrix-pipe time of the fp32 form.
 * `weight` = az_winograd_pack_filter_x3_f32 output (16-channel steps; source 1 starts on a step boundary); every other field,
 * the epilogue, split-K and gn_quads as az_conv2d_winograd_f32.  Inputs are assumed finite (an Inf operand splits into NaN
 * pieces).                                                                                                                  */
int az_conv2d_winograd_x3_f32(const AzConvArgs* args, az_stream_t stream);
/* fp32 operands on the fp16 matrix pipe ("f16x2"; AZ_FP32_MFMA=f16x2): HALF the matrix instructions of the bf16x3 form.
 * An activation x enters as x' = x * AZ_F16X2_IN_SCALE = h + l / 2^11 with h = fp16(x') and l = fp16((x' - h) * 2^11) (22 - 23
 * significant bits), a weight as w' = w * w_scale = wh + wl (two fp16 pieces, the residual unscaled) plus a third plane
 * whs = wh / 2^11, and a product is THREE partial products on v_mfma_f32_32x32x16_f16 with fp32 accumulation in ONE accumulator,
 *      wh h + wl h + whs l  =  w' x' - wl (x' - h)      (the dropped term is <= 2^-22 |w' x'|),
 * scaled back by 1 / (AZ_F16X2_IN_SCALE * w_scale) (powers of two: exact) before the fused epilogue.  Measured against fp64 the
 * result is as accurate as the fp32 MFMA kernel (the fp32 accumulation dominates both; tests/test_gpu_kernels.py::
 * test_conv2d_x3_accuracy).  The gfx950 matrix pipe honours fp16 subnormals (tools/mfma_denorm_probe.hip), so the pieces degrade
 * gracefully at the small end.  DOMAIN (stated, unlike bf16x3's, which is all of fp32): |x| * AZ_F16X2_IN_SCALE < 65520 --
 * activations up to ~1.0e6 (Winograd form: 4-pixel sums of the input, i.e. inputs up to ~2.6e5); beyond it a piece is Inf and the
 * outputs that depend on it are NaN (never a silently wrong finite value).  At the small end every operand carries an absolute
 * error <= 2^-36 / AZ_F16X2_IN_SCALE = 2.3e-10 (full relative precision from |x| >= 1e-3 on).  That range is for inputs whose
 * magnitude is bounded by construction (behind a normalisation); for any other input pass the sources' largest magnitude
 * (in_absmax0 / in_absmax1, from az_absmax_f32) and the kernel picks the scale itself: no stated range.  bf16x3 stays selectable
 * (AZ_FP32_MFMA=bf16x3).  `weight` = az_pack_conv_weight_f16x2_f32 / az_winograd_pack_filter_f16x2_f32 output (the layouts of the x3
 * packings: three 2-byte planes), `w_scale` the scale given to that packing; everything else as the x3 entries.  Same reference
 * op: azula/nn/layers.py:25-68 ConvNd / torch.nn.Linear on tokens (azula/nn/dit.py:88-93, azula/nn/attention.py:45-46). */
#define AZ_F16X2_IN_SCALE 0.0625f
int az_conv2d_f16x2_f32(const AzConvArgs* args, az_stream_t stream);
int az_conv2d_winograd_f16x2_f32(const AzConvArgs* args, az_stream_t stream);
/* The weight scale the f16x2 packings want for a weight tensor whose largest magnitude is `amax`: the power of two that puts
 * amax (times 2.25 for the Winograd transform G g G^T when `winograd`) into (2^13, 2^14]; 1 for amax = 0 / non-finite. */
float az_f16x2_weight_scale(float amax, int32_t winograd);
/* slots[i] = max |x| over the elements workgroup i visits, i < AZ_ABSMAX_SLOTS (every slot is written: no initialisation, no
 * atomics, deterministic): the largest magnitude of an activation tensor for AzConvArgs.in_absmax0 / in_absmax1.  One streaming
 * read of the tensor (HBM-bound).  NaN elements are ignored (they turn the outputs they reach into NaN through their own pieces). */
#define AZ_ABSMAX_SLOTS 256
int az_absmax_f32(float* slots, const float* x, int64_t n, az_stream_t stream);
/* The same slots WITHOUT reading the tensor, from the GroupNorm partial moments its producing convolution left (AzConvArgs.gn_quads:
 * `count` records of (n, mean, M2, -)): every element of a record satisfies |x| <= |mean| + sqrt(M2), so the slots hold an UPPER
 * bound of max |x| (loose by up to sqrt(n): a few binades of the f16x2 pieces' 30, no precision).  NaN moments are ignored. */
int az_absmax_from_moments_f32(float* slots, const float* partials, int64_t count, az_stream_t stream);
/* Winograd F(4x4,3x3) form (6x6 patches, 36 frequency GEMMs: 2.25 multiplies per output instead of 4 / 9).
 * NOT exact: the transforms multiply by 2, 4, 5, 8 and the filter transform by 1/4 .. 1/24, so the fp32
 * rounding error is ~20x that of the F(2x2) kernel (~1e-5 of the output scale per layer).  Opt-in
 * (AZ_WINOGRAD=4).  `weight` = az_winograd4_pack_filter_f32 output; other fields as az_conv2d_f32.  */
int az_conv2d_winograd4_f32(const AzConvArgs* args, az_stream_t stream);
int az_conv2d_winograd4_suggest_splitk(int64_t batch, int32_t hout, int32_t wout, int32_t cout_s, int32_t cin_s);
/* Suggested split-K factor for a conv shape on this device (pure function of the shape).     */
int az_conv2d_suggest_splitk(int64_t npix, int32_t cout_s, int32_t cin_s, int32_t ksize);
/* torch layout (cout, cin, ks, ks) -> packed [ks*ks][cout_s][cin_s] with zero padding; the
 * input channels [0, cin0) map to packed [0, cin0) and [cin0, cin) to [c0s, c0s + cin - cin0)
 * (two-source concat with padded strides).  Runs on the device.                               */
int az_pack_conv_weight_f32(float* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                            int32_t cin0, int32_t c0s, int32_t cin_s, az_stream_t stream);

/* ------------------------------------------------------------------ K4: multi-head self-attention
 * out[b, t, h, :] = softmax_s( scale * <q[b,t,h,:], k[b,s,h,:]> ) v[b,s,h,:], fp32 softmax, flash-style.
 * Element (b, t, h, c) of q is at q + b*q_bstride + t*q_tstride + h*q_hstride + c (c contiguous);
 * likewise k, v, out -- so one kernel serves the fused-QKV layouts of
 *   azula/nn/attention.py:89-104   '(n H C)' + per-head q/k RMSNorm (qk_rmsnorm = 1) + SDPA (scale = C^-1/2),
 *   plugins/adm/_src/unet.py:338-345  legacy '(H 3 C)' order, scale = C^-1/4 on q and k (pass C^-1/2),
 *   plugins/adm/_src/unet.py:371-379  new '(3 H C)' order,
 *   plugins/jit/_src/model.py:121-142  '(3 H C)' + weighted q/k RMSNorm + 2-D rotary + SDPA.
 * head_dim in {8 (fp32 only), 16, 32, 64, 80, 128} (80 = JiT-H); all strides multiples of 4 floats.                 */
typedef struct AzAttnArgs {
  const float* q;
  const float* k;
  const float* v;
  float* out;
  int32_t batch, heads, tokens, head_dim;
  int64_t q_bstride, q_tstride, q_hstride;
  int64_t k_bstride, k_tstride, k_hstride;
  int64_t v_bstride, v_tstride, v_hstride;
  int64_t o_bstride, o_tstride, o_hstride;
  float scale;
  int32_t qk_rmsnorm; /* 1: q and k rows are RMS-normalised (eps) before the dot product */
  float eps;
  /* channels the RMS norm averages over; 0 = head_dim.  The kernels exist for head_dim 16 / 32 / 64 / 80 / 128 (fp32: also 8):
   * the host runs any other head size d (azula/nn/attention.py:35-51 accepts every channels // attention_heads) by zero-padding
   * each head of q, k, v to the next instantiated size when it packs the projections -- zero channels change neither q.k nor
   * p.v -- and passes norm_dim = d here (scale is explicit: 1 / sqrt(d) of the REAL size).                                  */
  int32_t norm_dim;
  /* optional rotary embedding (azula/nn/attention.py:93-96,112-156): cos/sin of theta, laid out
   * (tokens, heads, head_dim / 2), shared by the batch; adjacent channel pairs (2i, 2i+1) of q and k are
   * rotated after the RMS norm.  NULL = no RoPE.                                                */
  const float* rope_cos;
  const float* rope_sin;
  /* optional learned gains (head_dim floats, shared by all heads) applied to the RMS-normalised q / k
   * before the rotation (plugins/jit/_src/model.py:108-109,129-133).  NULL = no gain.          */
  const float* q_weight;
  const float* k_weight;
  /* optional boolean attention mask (azula/nn/attention.py:72-104 -> scaled_dot_product_attention(attn_mask=mask)):
   * one byte per (query, key) pair, non-zero = the query may attend to the key; laid out (tokens, tokens) with a batch
   * and a head stride in bytes (0 = shared by all samples / heads).  A query whose keys are all masked yields NaN, as
   * in the reference.  NULL = no mask.                                                                            */
  const uint8_t* mask;
  int64_t mask_bstride, mask_hstride;
  /* az_attention_bf16_f32 / az_attention_f16_f32 only (0 elsewhere): 1 = q, k, v and out hold the entry's 2-byte type (the
   * activations of a module cast to half precision live in HBM in the module's type); strides stay in ELEMENTS, addresses and
   * strides need 8-byte granularity.  Norms, gains, RoPE and the softmax stay fp32.                                      */
  int32_t io_dtype;
  int32_t reserved;
} AzAttnArgs;
int az_attention_f32(const AzAttnArgs* args, az_stream_t stream);
/* The same operation for backbones cast to half precision: q / k / v / out stay fp32 tensors, norms, gains, RoPE and the
 * online softmax stay fp32, both contractions run on v_mfma_f32_32x32x16_{bf16,f16} (fp32 accumulate); exp via exp2. */
/* az_attention_f32's arithmetic on the bf16 matrix pipe at fp32 accuracy: q, k, v and the probabilities are split exactly into
 * three bf16 pieces and each product is six partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (softmax, norms,
 * RoPE in fp32); the attention of fp32 modules when AZ_FP32_MFMA = bf16x3 (the default).  head_dim 16, 32, 64, 80, 128.
 * Replaces the same reference lines as az_attention_f32. */
int az_attention_x3_f32(const AzAttnArgs* args, az_stream_t stream);
/* The same kernel in the f16x2 form (see az_conv2d_f16x2_f32): three partial products per contraction on v_mfma_f32_32x32x16_f16 --
 * the keys as [kh | kl | kh / 2^11] of k * 2^4 and the queries as [h | l] of q / 2^4 (the scales cancel); the probabilities (at most
 * 2^8 under the lazy running maximum) as three pieces of p * 2^6, the values as two of v / 2^4 (1 / 4 folded into the final 1 / l).
 * Softmax, norms, gains, RoPE in fp32 as before.  Domain: |k| < 4094, |v| and |q * scale * log2 e| < 1.0e6 (beyond: NaN).  Same
 * reference op: torch SDPA behind azula/nn/attention.py:89-104, plugins/adm/_src/unet.py:338-379.                          */
int az_attention_f16x2_f32(const AzAttnArgs* args, az_stream_t stream);
int az_attention_bf16_f32(const AzAttnArgs* args, az_stream_t stream);
int az_attention_f16_f32(const AzAttnArgs* args, az_stream_t stream);

/* y[r, c] = x[r, 2c] * silu(x[r, 2c+1]), c < cout (SwiGLU, azula/nn/layers.py:89-110); xs / ys = row strides. */
int az_swiglu_f32(float* y, const float* x, int64_t rows, int64_t cout, int64_t xs, int64_t ys, az_stream_t stream);

/* Token-window helpers for sequences that grow / shrink inside a backbone (JiT in-context class
 * tokens, plugins/jit/_src/model.py:362-374).  Token tensors are (B, tokens, cs), cs % 4 == 0.
 *   copy: dst[b, dst_off + i, :] = src[b, src_off + i, :]            i < n   (torch.cat / x[:, k:])
 *   fill: dst[b, dst_off + j, :] = row[b * row_bstride + :] + pos[j, :]  j < n
 *         (y_emb.unsqueeze(1).repeat(1, n, 1) + in_context_posemb)                                 */
int az_token_copy_f32(float* dst, int64_t dst_tokens, int64_t dst_off, const float* src, int64_t src_tokens,
                      int64_t src_off, int64_t n, int64_t B, int64_t cs, az_stream_t stream);
int az_token_fill_f32(float* dst, int64_t dst_tokens, int64_t dst_off, int64_t n, const float* row,
                      int64_t row_bstride, const float* pos, int64_t B, int64_t cs, az_stream_t stream);
/* fill with the DESTINATION in a 2-byte type (dtype 1: bfloat16, 2: IEEE half; row / pos fp32; cs % 8 == 0): the class tokens of a
 * module cast to half precision whose activations live in HBM in its own type.  (A copy between two such tensors is
 * az_token_copy_f32 with cs / 2: two 2-byte values per float.)                                                              */
int az_token_fill_h16(void* dst, int64_t dst_tokens, int64_t dst_off, int64_t n, const float* row, int64_t row_bstride,
                      const float* pos, int64_t B, int64_t cs, int32_t dtype, az_stream_t stream);

/* Sinusoidal timestep embedding (plugins/jit/_src/model.py:59-81): dst[r, 0:half] = cos(t_r f),
 * dst[r, half:2 half] = sin(t_r f), f_j = exp(-ln(max_period) j / half); t_r = t_dev[r * t_stride]
 * (t_stride 0: one device scalar, e.g. AzStepCoef.c_time of the current step).                     */
int az_timestep_embedding_f32(float* dst, int64_t ldd, const float* t_dev, int64_t t_stride, int64_t rows,
                              int32_t half, float max_period, az_stream_t stream);

/* Patchify NCHW (B, Z, H, W) -> tokens (B, H/p * W/p, cs), feature = z*p*p + a*p + b, scaled by
 * *scale_dev (NULL = 1), pad features zero; and back (azula/nn/layers.py:198-247, azula/nn/vit.py:92-106). */
int az_patchify_f32(float* dst, const float* src, const float* scale_dev, int64_t B, int64_t Z, int64_t H, int64_t W,
                    int64_t p, int64_t cs, az_stream_t stream);
int az_unpatchify_f32(float* dst, const float* src, int64_t B, int64_t Z, int64_t H, int64_t W, int64_t p, int64_t cs,
                      az_stream_t stream);

/* az_pack_conv_weight_f32's layout in 2-byte elements (f16 != 0: IEEE half, else bfloat16; round to nearest even). */
int az_pack_conv_weight_half_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                                 int32_t cin0, int32_t c0s, int32_t cin_s, int32_t f16, az_stream_t stream);
/* The same layout as three bf16 planes [piece][tap][cout_s][cin_s] with w = w1 + w2 + w3 exactly (az_conv2d_x3_f32);
 * dst holds 3 * ks*ks*cout_s*cin_s two-byte elements.                                                              */
int az_pack_conv_weight_x3_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                               int32_t cin0, int32_t c0s, int32_t cin_s, az_stream_t stream);
/* The same three-plane layout for az_conv2d_f16x2_f32: planes [wh | wl | wh / 2^11] of w' = w * w_scale as IEEE half values
 * (wh = fp16(w'), wl = fp16(w' - wh)); w_scale = az_f16x2_weight_scale(max |w|, 0) (a power of two).                    */
int az_pack_conv_weight_f16x2_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t ks, int32_t cout_s,
                                  int32_t cin0, int32_t c0s, int32_t cin_s, float w_scale, az_stream_t stream);

/* torch (cout, cin, 3, 3) -> Winograd filter transform U = G g G^T (fp64 accumulate, one rounding) in
 * the layout az_conv2d_winograd_f32 streams: [nk chunks of 8 cin][cblocks of 64 cout][16][64][8]; input
 * channels [0, cin0) fill chunks [0, nk0), the rest start at chunk nk0 (two-source concat).      */
int az_winograd_pack_filter_f32(float* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t nk0,
                                int32_t nk, int32_t cblocks, az_stream_t stream);

/* The same transform for az_conv2d_winograd_x3_f32: U (rounded to fp32 as above) split exactly into three bf16 pieces, in the
 * order that kernel's waves load it as MFMA A fragments: [nk steps of 16 cin][cblocks of 64 cout][wave 0..7 = k * 4 + frequency
 * row xi][f: nu = k + 2 f][32-cout half][piece][lane][8 bf16] (96 KB per step and cout block; opaque to callers).  dst holds
 * nk * cblocks * 49152 two-byte elements; channels [0, cin0) fill steps [0, nk0), the rest start at step nk0.      */
int az_winograd_pack_filter_x3_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t nk0,
                                   int32_t nk, int32_t cblocks, az_stream_t stream);
/* The same fragment-ordered layout for az_conv2d_winograd_f16x2_f32: the three 2-byte pieces of an element are
 * [uh | ul | uh / 2^11] of u' = U * w_scale (IEEE half), w_scale = az_f16x2_weight_scale(max |g|, 1).                    */
int az_winograd_pack_filter_f16x2_f32(void* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t nk0,
                                      int32_t nk, int32_t cblocks, float w_scale, az_stream_t stream);

/* Same for F(4x4,3x3) (points 0, +-1, +-2, inf): [nk chunks of 4 cin][cblocks of 64 cout][36][64][4]; packed
 * channel position pc maps to input channel pc (pc < cin0) or cin0 + pc - c0s (pc >= c0s).        */
int az_winograd4_pack_filter_f32(float* dst, const float* src, int32_t cout, int32_t cin, int32_t cin0, int32_t c0s,
                                 int32_t nk, int32_t cblocks, az_stream_t stream);

/* ------------------------------------------------------------------ hipGraph helpers (host side)
 * Capture everything enqueued on `stream` between begin/end into an executable graph.         */
typedef struct AzGraph AzGraph;
int az_graph_begin(az_stream_t stream);
int az_graph_end(az_stream_t stream, AzGraph** out);
int az_graph_launch(AzGraph* g, az_stream_t stream);
int az_graph_destroy(AzGraph* g);
int az_graph_num_nodes(AzGraph* g, int64_t* n);

/* ------------------------------------------------------------------ fp64 latents
 * `Sampler(dtype=torch.float64)`: the reference's schedule scalars become fp64 tensors of shape (1, ..., 1) and promote
 * the latents to fp64 in the preconditioning (azula/denoise.py:306-322) and in every sampler update
 * (azula/sample.py:210-214, 257-259, 296-303, 343-350, 417-431, 975-993), while the backbone keeps its dtype.
 *   az_transition_f64:   the flat form of az_transition_f32 on fp64 tensors (x_t, F = posterior mean or backbone output,
 *                        eps, x_s, mean_out are double*; `coef` points to 12 doubles in AzStepCoef's field order)
 *   az_axpby_f64:        y = a x + b z, fp64, z read as float (z_is_f32) or double; a, b device scalars or per-row
 *   az_scale_f64_to_f32: y = (float)(s x)   -- `(c_in * x_t).to(backbone dtype)`, azula/denoise.py:317                */
int az_transition_f64(const AzTransitionArgs* args, az_stream_t stream);
/* Captured loop with fp64 latents (round 4): cur[0, words) = table[*step_counter - 1][0, words) -- the fp64 coefficients of the
 * step az_step_begin has just started (the fp32 row feeds the backbone's time embedding, this one az_scale_f64_to_f32 /
 * az_axpby_f64 / az_transition_f64 through pointers into `cur`), so that ONE graph serves every step.  words <= 64. */
int az_step_row_f64(double* cur, const double* table, const int32_t* step_counter, int32_t n_rows, int32_t words,
                    az_stream_t stream);
int az_axpby_f64(double* y, const double* a_dev, const double* x, const double* b_dev, const void* z, int32_t z_is_f32,
                 int64_t rows, int64_t inner, int32_t a_stride, az_stream_t stream);
int az_scale_f64_to_f32(float* y, const double* x, const double* s_dev, int64_t rows, int64_t inner, int32_t s_stride,
                        az_stream_t stream);

/* ------------------------------------------------------------------ multi-GPU sampling (no reference counterpart:
 * the reference draws `torch.randn_like(x_t)` of the whole batch on one device, azula/sample.py:214,259)
 * dst[e] = element start + e of the standard-normal tensor that torch.randn / Tensor.normal_ (float32) would draw with the
 * Philox state (seed, offset) in a launch of `threads_total` threads (ATen's policy: 256 x min(ceil(numel / 256),
 * CUs x max resident threads / 256)) -- bit for bit, so that a rank of a batch-sharded run draws only ITS samples of the
 * single-device random stream.  The caller advances the generator's offset as the full draw would.                        */
int az_randn_slice_f32(float* dst, uint64_t seed, uint64_t offset, int64_t threads_total, int64_t start, int64_t count,
                       az_stream_t stream);

/* ------------------------------------------------------------------ measurement support (not on the sampling path)
 * Known-traffic kernels that calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 (MI355X_MICROARCH.md,
 * section HBM: "calibrate on a known byte count in your own access pattern").  `az_calib_read_f32` reads every byte of
 * `src[0, nbytes)` exactly once: the buffer is rows of `row_bytes`, each read as sectors of `group_bytes` contiguous
 * bytes by `group_bytes / width` adjacent lanes (`width` = 8 or 16 bytes per lane), the sectors of a row by consecutive
 * instructions.  `az_calib_write_f32` fills `dst[0, nbytes)` with 16-byte-per-lane stores.  The reference has no
 * counterpart (it has no native code); used by tools/pmc_traffic.py and bench.py's roofline leg only.               */
int az_calib_read_f32(const float* src, float* sink, int64_t nbytes, int32_t width, int32_t group_bytes,
                      int64_t row_bytes, az_stream_t stream);
int az_calib_write_f32(float* dst, int64_t nbytes, float value, az_stream_t stream);
/* `workgroups` x 4 waves, each issuing iters x 8 independent v_mfma_f32_32x32x2_f32 on registers (4096 FLOP each): the fp32
 * MFMA rate the chip SUSTAINS under its power limit, which bench.py reports beside the nominal 157.3 TF/s peak. */
int az_calib_mfma_f32(float* sink, int32_t workgroups, int32_t iters, float a, float b, az_stream_t stream);
/* The same instruction stream on per-lane PSEUDO-RANDOM operands (uniform in [-a, a) x [-b, b), a different pair per
 * accumulator chain).  `az_calib_mfma_f32` multiplies the same two constants forever, which toggles almost nothing (~690 W on
 * MI355X); on random bit patterns the matrix pipe alone draws what a real GEMM's does, and the 1400 W cap -- not the 2.4 GHz
 * clock -- sets the rate (tools/power_probe.py).  bench.py reports both as the context of `roofline.frac`. */
int az_calib_mfma_random_f32(float* sink, int32_t workgroups, int32_t iters, float a, float b, az_stream_t stream);
/* The bf16 pipe: 8 independent v_mfma_f32_32x32x16_bf16 per iteration (32768 FLOP each) on per-lane random bf16 operands:
 * the rate the power cap leaves the instruction of the bf16x3 kernels (nominal 2516.8 TF/s). */
int az_calib_mfma_random_bf16(float* sink, int32_t workgroups, int32_t iters, float a, float b, az_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AZULA_AMD_H */
