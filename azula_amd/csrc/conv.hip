// K3 / K5: implicit-GEMM convolution and token-linear on gfx950 fp32 MFMA.
//
//   out[pixel, co] = sum_{tap, ci} W[tap][co][ci] * in[pixel + tap, ci]
//
// GEMM view: "A" = packed weights (rows = output channels), "B" = gathered input pixels
// (columns = output pixels), K = taps x input channels, accumulated with
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain; 157.3 TFLOP/s chip peak, same as the VALU
// peak, but one operand VGPR per 2048 MACs instead of two per MAC).  Weights are the A operand
// so that each lane ends up holding 4 CONSECUTIVE output channels of one pixel: the NHWC
// epilogue (bias, SiLU, gate, residual) and the store are all 16-byte vectors.
//
// Block tile 128 (co) x 128 (pixels) x 32 (k); 4 waves in 2 x 2, each 64 x 64 = 2 x 2 MFMA
// tiles (64 accumulator VGPRs).  LDS: both operand tiles are stored [row][k] with a 36-float
// row stride: ds_write_b128 from the loader and ds_read_b128 fragment reads are both
// bank-conflict-free (36*r mod 64 visits 16 distinct 4-bank slots).  Double-buffered LDS with
// register staging: the global loads of tile t+1 are issued before the MFMAs of tile t and
// written to the other buffer after them (one barrier per K-tile); 2 blocks/CU co-reside so
// one block's MFMAs cover the other's barrier/load latency.
//
// The gather handles zero padding, stride, two concatenated sources and read-side nearest x2
// upsampling, so concat / upsample / pad never touch HBM as separate passes.  It is built on
// bounds-checked buffer loads: a per-lane byte offset that only changes when the (tap, source)
// changes (out-of-range = reads as zero, so padding / ragged tiles / channel tails need no
// branches) plus a wave-uniform scalar offset that walks K.
//
// This file holds two kernels behind the same AzConvArgs: the direct implicit GEMM below (any
// ksize / stride, linears) and the fused Winograd F(2x2,3x3) kernel further down (3x3 stride 1).
#include "conv_shared.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 128;   // output-channel tile
constexpr int BN = 128;   // pixel tile
constexpr int BK = 32;    // k tile (input channels of one tap)
constexpr int LDSS = 36;  // padded LDS row stride in floats (144 B, 16-B aligned)
constexpr int TILE_F = (BM + BN) * LDSS;

#include "igemm_kloop.inc"  // (A/B variants of the stream: tools/kloop_variant.py builds from a patched COPY of this directory)
struct ConvP {
  AzConvArgs a;
  int npix;     // batch * hout * wout
  int cin_s;    // c0s + c1s
  int nkc0;     // K-tiles of source 0 per tap: ceil(c0s / BK)
  int nkc1;     // K-tiles of source 1 per tap
  int nk;       // taps * (nkc0 + nkc1)
  int kps;      // K-tiles per split
  int tiles_m;  // ceil(cout_s / BM)
  int tiles_n;  // ceil(npix / BN)
  int asm_loop; // fp32 K-32 kernel: one tap, whole K tiles -> the hand-scheduled K loop (igemm_kloop.inc)
  int m_tile0;  // bf16x3 kernel: first output-channel tile of this launch (the 256 x 256 kernel took the tiles before it)
  float out_scale;  // f16x2 kernels: 1 / w_scale; the accumulators are multiplied by out_scale / (the activation scale) behind the K loop (powers of two)
};


// Epilogue of a 2 x 2 arrangement of 32x32 MFMA accumulators (wave sub-tile 64 couts x 64 pixels): the lane holds,
// for pixel (lane & 31) of each pixel tile, channels ct*32 + 8*q + 4*(lane >> 5) + [0, 4) in registers 4q .. 4q+3
// (the C/D layout of the 32x32 MFMAs is the same for f32, bf16 and f16 operands).  Stored straight from that layout an
// instruction writes 32 bytes of 32 different pixels; instead the block's 128 x 128 outputs go through LDS (`obuf`,
// the operand stages, free after the K loop's last barrier) as [pixel][cout] rows (stride padded by 4 floats:
// conflict-free ds_write_b128) and are read back so that 32 consecutive lanes store the 512 contiguous bytes of one
// pixel.  The planar (NCHW) destination keeps the direct form, whose lanes run along pixels.
constexpr int OSTR = BM + 4;  // floats per pixel row of the exchange buffer
static_assert((BN * OSTR + BN) * 4 <= 2 * TILE_F * 4, "epilogue exchange buffer must fit in the operand stages");

template <int IO = 0>  // element type of dst / res (conv_shared.h: ld4_io / st4_io); the planar destination is always fp32
__device__ __forceinline__ void store_acc_tiles(const ConvP& p, const f32x16 (&acc)[2][2], int m0, int n0, int wc, int wp,
                                                int lane, float* obuf) {
  const AzConvArgs& a = p.a;
  if (a.dst_nchw && a.splitk == 1) {
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int n = n0 + wp * 64 + pt * 32 + (lane & 31);
      if (n >= p.npix) continue;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = m0 + wc * 64 + ct * 32 + 8 * q + 4 * (lane >> 5);
          if (co >= a.cout_s) continue;
          epilogue_store(a, n, co, make_float4(acc[ct][pt][4 * q + 0], acc[ct][pt][4 * q + 1], acc[ct][pt][4 * q + 2],
                                               acc[ct][pt][4 * q + 3]));
        }
      }
    }
    return;
  }
  const int tid = threadIdx.x;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    float* orow = obuf + (wp * 64 + pt * 32 + (lane & 31)) * OSTR + wc * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(orow + ct * 32 + 8 * q) =
            make_float4(acc[ct][pt][4 * q + 0], acc[ct][pt][4 * q + 1], acc[ct][pt][4 * q + 2], acc[ct][pt][4 * q + 3]);
  }
  int* pimg = reinterpret_cast<int*>(obuf + BN * OSTR);  // image index of each pixel of the tile (-1: past the end)
  if (tid < BN) {
    const int n = n0 + tid;
    pimg[tid] = n < p.npix ? n / (a.hout * a.wout) : -1;
  }
  __syncthreads();
  const int cq = tid & (BM / 4 - 1);  // the same channel quad in every iteration (256 % (BM / 4) == 0)
  const int co = m0 + cq * 4;
  if (co >= a.cout_s) return;
  constexpr int NIT = BN * (BM / 4) / 256, NB = 8;
#pragma unroll 1
  for (int it0 = 0; it0 < NIT; it0 += NB) {
    int n[NB], b[NB];
    float4 v[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int px = ((it0 + i) * 256 + tid) / (BM / 4);
      b[i] = pimg[px];
      n[i] = b[i] >= 0 ? n0 + px : -1;
      v[i] = *reinterpret_cast<const float4*>(obuf + px * OSTR + cq * 4);
    }
    epilogue_store_batch<NB, false, IO>(a, n, b, co, v, (int64_t)blockIdx.y * p.npix);
  }
}

// The same exchange in two halves of 64 pixels (34 KB instead of 68 KB of LDS), for the K-tile-16 instantiation whose
// operand stages are too small for the whole 128 x 128 tile: the waves of pixel half `hh` park their accumulators,
// all four waves store.  (NCHW destinations take the direct path of store_acc_tiles: no LDS.)
constexpr int OBUF_HALF_F = 64 * OSTR + 64;
__device__ __forceinline__ void store_acc_tiles_halves(const ConvP& p, const f32x16 (&acc)[2][2], int m0, int n0, int wc, int wp,
                                                        int lane, float* obuf) {
  const AzConvArgs& a = p.a;
  if (a.dst_nchw && a.splitk == 1) {
    store_acc_tiles(p, acc, m0, n0, wc, wp, lane, obuf);
    return;
  }
  const int tid = threadIdx.x;
  const int cq = tid & (BM / 4 - 1);
  const int co = m0 + cq * 4;
  int* pimg = reinterpret_cast<int*>(obuf + 64 * OSTR);
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    if (wp == hh) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        float* orow = obuf + (pt * 32 + (lane & 31)) * OSTR + wc * 64 + 4 * (lane >> 5);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(orow + ct * 32 + 8 * q) =
                make_float4(acc[ct][pt][4 * q + 0], acc[ct][pt][4 * q + 1], acc[ct][pt][4 * q + 2], acc[ct][pt][4 * q + 3]);
      }
    }
    if (tid < 64) {
      const int n = n0 + hh * 64 + tid;
      pimg[tid] = n < p.npix ? n / (a.hout * a.wout) : -1;
    }
    __syncthreads();
    if (co < a.cout_s) {
      // 8 outputs per thread in two batches of 4: the waves of the second pixel half still hold their 64 accumulator
      // registers here, and with a batch of 8 (values + gate + residual + indices) the 168-register budget of three
      // workgroups per CU overflowed into scratch (85 spilled registers in round 2's build)
      constexpr int NB = 4;
#pragma unroll 1
      for (int i0 = 0; i0 < 64 * (BM / 4) / 256; i0 += NB) {
        int n[NB], b[NB];
        float4 v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int px = ((i0 + i) * 256 + tid) / (BM / 4);
          b[i] = pimg[px];
          n[i] = b[i] >= 0 ? n0 + hh * 64 + px : -1;
          v[i] = *reinterpret_cast<const float4*>(obuf + px * OSTR + cq * 4);
        }
        epilogue_store_batch<NB>(a, n, b, co, v, (int64_t)blockIdx.y * p.npix);
      }
    }
    __syncthreads();
  }
}

// KT = K tile (channels of one tap per stage).  32: 73.7 KB of LDS, two workgroups per CU.  16: 41 KB, THREE per CU --
// chosen by the host when the tile count quantises badly over 2 x 256 slots (e.g. the 16384 x 768 -> 768 token GEMM:
// 768 tiles = 3 per CU, i.e. a third of the time one lone workgroup per CU, which cannot keep the matrix pipe full).
template <int KT>
__global__ __launch_bounds__(256, KT == 32 ? 2 : 3) void conv_igemm_kernel(ConvP p) {
  constexpr int LS = KT + 4;                // padded LDS row stride (floats): conflict-free ds_read_b128 fragments
  constexpr int TF = (BM + BN) * LS;        // floats per stage
  constexpr int CPR = KT / 4;               // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;            // rows per loader pass
  constexpr int NP = 128 / RPP;             // loader passes (per operand)
  constexpr int SMEM_F = 2 * TF > (KT == 32 ? BN * OSTR + BN : OBUF_HALF_F) ? 2 * TF : (KT == 32 ? BN * OSTR + BN : OBUF_HALF_F);
  __shared__ __attribute__((aligned(16))) float smem[SMEM_F];
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wc = wid >> 1;  // wave's 64-channel half
  const int wp = wid & 1;   // wave's 64-pixel half

  // XCD-aware bijective remap: each XCD (private L2) walks a contiguous range of tiles, and
  // consecutive tiles share the same pixel tile (activation rows), differing in channels.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tile_n = wg / p.tiles_m;
  const int tile_m = wg - tile_n * p.tiles_m;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);

  // ---- loader coordinates: thread -> (16-B chunk cc along k, rows r0 + 32*i)
  const int cc = tid % CPR;
  const int r0 = tid / CPR;
  const int hw_out = a.hout * a.wout;
  const int b_first = n0 / hw_out;  // wave-uniform: offsets are relative to this sample

  // Buffer descriptors (wave-uniform: kernel arguments + blockIdx only).
  const int64_t s0_elems = (int64_t)a.h0 * a.w0 * a.c0s;
  // (depth taps: per-lane source planes are taken relative to b_base; lanes whose plane is masked may land anywhere)
  const int b_base = az_depth_base(a, b_first);
  const int64_t s1_elems = (int64_t)a.h1 * a.w1 * a.c1s;
  auto clamp_bytes = [](int64_t e) { return (unsigned)(e <= 0 ? 0 : (e * 4 > AZ_RSRC_CLAMP ? AZ_RSRC_CLAMP : e * 4)); };  // (e <= 0: a depth shift past the last plane)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.weight, 0, clamp_bytes((int64_t)a.ksize * a.ksize * a.cout_s * p.cin_s), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src0 + b_base * s0_elems), 0, clamp_bytes((a.batch - b_base) * s0_elems), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 + b_base * s1_elems : a.src0), 0,
      a.src1 ? clamp_bytes((a.batch - b_base) * s1_elems) : 0u, 0x00020000);

  // Per-thread constants: weight-row byte offsets, pixel coordinates.
  unsigned voffW[NP];
  int prel[NP], ihb[NP], iwb[NP];
  bool pdok[NP];  // AzConvArgs.depth: the source plane (image + depth_shift) lies inside the image's volume
  int psrc[NP];   // source plane of the pixel's image, relative to b_base (== prel without depth taps)
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int co = m0 + r0 + RPP * i;
    voffW[i] = co < a.cout_s ? (unsigned)((co * p.cin_s + cc * 4) * 4) : OOB;
    const int n = n0 + r0 + RPP * i;
    const bool pv = n < p.npix;
    const int nn = pv ? n : 0;
    const int b = nn / hw_out;
    const int rem = nn - b * hw_out;
    const int oh = rem / a.wout;
    const int ow = rem - oh * a.wout;
    prel[i] = pv ? b - b_first : -1;
    psrc[i] = az_depth_plane(a, b, pdok[i]) - b_base;
    ihb[i] = oh * a.stride - a.pad;
    iwb[i] = ow * (a.aniso ? a.stride_w : a.stride) - a.pad;
  }

  // K iterator (wave-uniform): tap -> source -> 32-channel chunk.  The per-lane activation
  // offsets voffA change only when (tap, source) changes; inside, the K walk is a scalar offset.
  const int nk_tap = p.nkc0 + p.nkc1;
  int it_tap = kt_begin / nk_tap;
  int it_r = kt_begin - it_tap * nk_tap;
  int it_src = it_r >= p.nkc0 ? 1 : 0;
  int it_kc = it_src ? it_r - p.nkc0 : it_r;
  unsigned voffA[NP];

  auto set_tap_src = [&]() {
    const int ky = it_tap / a.ksize;
    const int kx = it_tap - ky * a.ksize;
    const int cs = it_src ? a.c1s : a.c0s;
    const int up = it_src ? a.up1 : a.up0;
    const int upw = a.aniso ? (it_src ? a.up1_w : a.up0_w) : up;
    const int hs = it_src ? a.h1 : a.h0;
    const int ws = it_src ? a.w1 : a.w0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int ih = wrap_coord(ihb[i] + ky, a.hin, a.pad_mode);
      const int iw = wrap_coord(iwb[i] + kx, a.win, a.pad_mode);
      const bool ok = prel[i] >= 0 && pdok[i] && (unsigned)ih < (unsigned)a.hin && (unsigned)iw < (unsigned)a.win;
      const int pix = (psrc[i] * hs + (ih >> up)) * ws + (iw >> upw);
      voffA[i] = ok ? (unsigned)((pix * cs + cc * 4) * 4) : OOB;
    }
  };

  float4 ra[NP], rb[NP];

  auto load_tile = [&]() {
    const int cs = it_src ? a.c1s : a.c0s;
    const int kbase = it_kc * KT;                        // channel offset inside the source
    const int kglob = (it_src ? a.c0s : 0) + kbase;      // channel offset in the packed weights
    const unsigned soffW = (unsigned)(((int64_t)it_tap * a.cout_s * p.cin_s + kglob) * 4);
    const unsigned soffA = (unsigned)(kbase * 4);
    const bool kv = kbase + cc * 4 < cs;  // channel tail of this source (lane-dependent only there)
#pragma unroll
    for (int i = 0; i < NP; ++i) ra[i] = buf_ld4(rw, voffW[i], soffW);
    if (it_src) {
#pragma unroll
      for (int i = 0; i < NP; ++i) rb[i] = buf_ld4(rs1, kv ? voffA[i] : OOB, soffA);
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) rb[i] = buf_ld4(rs0, kv ? voffA[i] : OOB, soffA);
    }
  };

  auto advance = [&]() {
    ++it_kc;
    if (it_kc == (it_src ? p.nkc1 : p.nkc0)) {
      it_kc = 0;
      ++it_src;
      if (it_src == 2 || p.nkc1 == 0) {
        it_src = 0;
        ++it_tap;
      }
      set_tap_src();
    }
  };

  auto store_tile = [&](int buf) {
    float* As = smem + buf * TF;
    float* Bs = As + BM * LS;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      *reinterpret_cast<float4*>(As + (r0 + RPP * i) * LS + cc * 4) = ra[i];
      *reinterpret_cast<float4*>(Bs + (r0 + RPP * i) * LS + cc * 4) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_off = (lane & 31) * LS + (lane >> 5) * 4;

  bool done = false;
  if constexpr (KT == 32) {
    // ONE tap, whole 32-channel K tiles (1 x 1 convolutions / token linears): the K loop as the hand-scheduled stream of
    // igemm_kloop.inc (gen_igemm_kloop.py) -- the same stage layout, loader coordinates and fragments as the C++ loop below
    if (p.asm_loop && kt_begin < kt_end) {
      typedef __attribute__((address_space(3))) float lds_float;
      const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)smem;
      const unsigned fa = lds0 + (unsigned)((wc * 64) * LS + frag_off) * 4u;
      const unsigned fb = lds0 + (unsigned)(BM * LS + (wp * 64) * LS + frag_off) * 4u;
      const unsigned stA = lds0 + (unsigned)(r0 * LS + cc * 4) * 4u;
      const unsigned dA = fa ^ (fa + TF * 4), dB = fb ^ (fb + TF * 4), dS = stA ^ (stA + TF * 4);
      const bool start1 = p.asm_loop == 1 && kt_begin >= p.nkc0;  // (one tap: the slice starts in the second source)
      const int kt_switch = (!start1 && kt_end > p.nkc0) ? p.nkc0 : 0x7fffffff;
      unsigned vA0[NP], vA1[NP];
      it_tap = 0;
      it_src = start1 ? 1 : 0;
      set_tap_src();
#pragma unroll
      for (int i = 0; i < NP; ++i) vA0[i] = voffA[i];
      it_src = 1;
      if (kt_switch != 0x7fffffff) set_tap_src();
#pragma unroll
      for (int i = 0; i < NP; ++i) vA1[i] = voffA[i];
      const uint64_t bw = (uint64_t)(uintptr_t)a.weight;
      const uint64_t b0 = (uint64_t)(uintptr_t)(a.src0 + b_base * s0_elems);
      const uint64_t b1 = (uint64_t)(uintptr_t)(a.src1 ? a.src1 + b_base * s1_elems : a.src0);
      const unsigned nw_ = clamp_bytes((int64_t)a.ksize * a.ksize * a.cout_s * p.cin_s);
      const unsigned n0_ = clamp_bytes((a.batch - b_base) * s0_elems), n1_ = a.src1 ? clamp_bytes((a.batch - b_base) * s1_elems) : 0u;
      unsigned f0 = (unsigned)b0, f1 = (unsigned)(b0 >> 32) & 0xffffu, f2 = n0_;
      const unsigned g0 = (unsigned)b1, g1 = (unsigned)(b1 >> 32) & 0xffffu, g2 = n1_;
      if (start1) f0 = g0, f1 = g1, f2 = g2;
      const unsigned dflags = 0x00020000u;
      const int nst = kt_end - kt_begin;
      const unsigned soffW0 = (unsigned)kt_begin * (KT * 4), soffA0 = (unsigned)(start1 ? kt_begin - p.nkc0 : kt_begin) * (KT * 4);
      if (p.asm_loop == 2) {
        // several taps (3 x 3 ... convolutions at any stride), ONE source, no upsampling, zero padding: a pixel row's offset is
        // linear in the tap -- base (window corner) + (ky W + kx) cs 4 -- wherever the tap is inside the image; a bit mask per
        // row says where (the stream turns the others into the out-of-bounds offset at every tap change)
        const int ks = a.ksize, tap0 = kt_begin / p.nkc0, kc0 = kt_begin - tap0 * p.nkc0;
        const int ky0 = tap0 / ks, kx0 = tap0 - ky0 * ks;
        unsigned pbase[NP], vmask[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          pbase[i] = (unsigned)((((psrc[i] * a.h0 + ihb[i]) * a.w0 + iwb[i]) * a.c0s + cc * 4) * 4);
          unsigned m = 0;
          for (int t = 0; t < ks * ks; ++t) {
            const int ih = ihb[i] + t / ks, iw = iwb[i] + t % ks;
            if (prel[i] >= 0 && pdok[i] && (unsigned)ih < (unsigned)a.hin && (unsigned)iw < (unsigned)a.win) m |= 1u << t;
          }
          vmask[i] = m;
        }
        const unsigned tap_kt = (unsigned)p.nkc0;
        const unsigned wadj = (unsigned)(a.cout_s * p.cin_s * 4 - p.nkc0 * (KT * 4));
        const unsigned cs4 = (unsigned)(a.c0s * 4), row_adj = (unsigned)((a.w0 - ks) * a.c0s * 4);
        const unsigned packed = (unsigned)ks | (unsigned)tap0 << 8 | (unsigned)kx0 << 16 | (unsigned)ky0 << 24;
        const unsigned soffW0t = (unsigned)(((int64_t)tap0 * a.cout_s * p.cin_s + kc0 * KT) * 4), soffA0t = (unsigned)(kc0 * (KT * 4));
        asm volatile(IGEMM_KLOOP_TAPS_ASM
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])
                     : "v"(fa), "v"(fb), "v"(stA), "v"(dA), "v"(dB), "v"(dS), "v"(voffW[0]), "v"(voffW[1]), "v"(voffW[2]), "v"(voffW[3]),
                       "v"(pbase[0]), "v"(pbase[1]), "v"(pbase[2]), "v"(pbase[3]), "v"(vmask[0]), "v"(vmask[1]), "v"(vmask[2]), "v"(vmask[3]),
                       "s"((unsigned)bw), "s"((unsigned)(bw >> 32) & 0xffffu), "s"(nw_), "s"(dflags), "s"(f0), "s"(f1), "s"(f2), "s"(dflags),
                       "s"(tap_kt), "s"(wadj), "s"(cs4), "s"(row_adj), "s"(nst), "s"(kt_begin), "s"(packed), "s"(soffW0t), "s"(soffA0t)
                     : IGEMM_KLOOP_TAPS_CLOBBERS);
      } else {
        asm volatile(IGEMM_KLOOP_ASM
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])
                     : "v"(fa), "v"(fb), "v"(stA), "v"(dA), "v"(dB), "v"(dS), "v"(voffW[0]), "v"(voffW[1]), "v"(voffW[2]), "v"(voffW[3]),
                       "v"(vA0[0]), "v"(vA0[1]), "v"(vA0[2]), "v"(vA0[3]), "v"(vA1[0]), "v"(vA1[1]), "v"(vA1[2]), "v"(vA1[3]),
                       "s"((unsigned)bw), "s"((unsigned)(bw >> 32) & 0xffffu), "s"(nw_), "s"(dflags), "s"(f0), "s"(f1), "s"(f2), "s"(dflags),
                       "s"(g0), "s"(g1), "s"(g2), "s"(dflags), "s"(nst), "s"(kt_begin), "s"(kt_switch), "s"(soffW0), "s"(soffA0)
                     : IGEMM_KLOOP_CLOBBERS);
      }
      done = true;
    }
  }

  if (!done) {
  if (kt_begin < kt_end) {
    set_tap_src();
    load_tile();
    store_tile(0);
  }
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) {
      advance();
      load_tile();  // global loads in flight under the MFMAs below
    }

    const float* As = smem + buf * TF + (wc * 64) * LS + frag_off;
    const float* Bs = smem + buf * TF + BM * LS + (wp * 64) * LS + frag_off;
#pragma unroll
    for (int kk = 0; kk < KT / 8; ++kk) {
      const float4 a0 = ld4(As + kk * 8);
      const float4 a1 = ld4(As + 32 * LS + kk * 8);
      const float4 b0 = ld4(Bs + kk * 8);
      const float4 b1 = ld4(Bs + 32 * LS + kk * 8);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w};
      const float av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w};
      const float bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  }

  // (thread coordinates of the epilogue from an opaque copy of the thread index: nothing derived from it has to survive the
  // asm statement, whose clobbers leave the compiler v0..v151)
  int te = threadIdx.x;
  asm volatile("" : "+v"(te));
  const int lane_e = te & 63, wc_e = te >> 7, wp_e = (te >> 6) & 1;
  if constexpr (KT == 32) store_acc_tiles(p, acc, m0, n0, wc_e, wp_e, lane_e, smem);
  else store_acc_tiles_halves(p, acc, m0, n0, wc_e, wp_e, lane_e, smem);
}

// =================================================================================================
// Half-precision-operand variant of the direct kernel, for backbones whose module was cast with .bfloat16() /
// .half() (the reference's mixed-precision route, azula/denoise.py:314-320): weights are packed once as bf16 / f16,
// activations stay fp32 in HBM and are rounded to the operand type while they are staged to LDS, products are
// accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16} (16x the fp32 MFMA rate) and the epilogue is the fp32 one.
// Same tile (128 couts x 128 pixels), K tile = 64 channels of one tap; LDS rows of 64 two-byte values with a 144-byte
// stride (= the fp32 kernel's 36-dword stride: ds_read_b128 fragments stay conflict-free); the accumulator layout of
// the 32x32 MFMAs is dtype-independent, so the store code is shared with conv_igemm_kernel.
constexpr int HBK = 64;                 // k tile (input channels of one tap)
constexpr int HLDS = 72;                // LDS row stride in 2-byte elements (144 B)
constexpr int HTILE = (BM + BN) * HLDS; // 2-byte elements per stage

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// SRCH: the sources hold the operand type already (AzConvArgs.src_dtype = 1: half-precision activations in HBM) -- a row's 64
// k-values are 128 bytes, thread -> (16-byte chunk = 8 values, rows (tid >> 3) + 32 i): half the loads, no conversion.
template <bool F16, bool SRCH = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_half_kernel(ConvP p) {
  constexpr int ES = SRCH ? 2 : 4;         // bytes per source element
  constexpr int NA = SRCH ? 4 : 8;         // activation loads per thread and stage
  constexpr int ACH = SRCH ? 8 : 4;        // k-values per 16-byte activation chunk
  __shared__ __attribute__((aligned(16))) unsigned short hsm[2 * HTILE];
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wc = wid >> 1;
  const int wp = wid & 1;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tile_n = wg / p.tiles_m;
  const int tile_m = wg - tile_n * p.tiles_m;
  const int m0 = (p.m_tile0 + tile_m) * BM;
  const int n0 = tile_n * BN;
  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);

  const int hw_out = a.hout * a.wout;
  const int b_first = n0 / hw_out;
  const int64_t s0_elems = (int64_t)a.h0 * a.w0 * a.c0s;
  const int64_t s1_elems = (int64_t)a.h1 * a.w1 * a.c1s;
  auto clamp_bytes = [](int64_t e) { return (unsigned)(e <= 0 ? 0 : (e > AZ_RSRC_CLAMP ? AZ_RSRC_CLAMP : e)); };
  const int b_base = az_depth_base(a, b_first);  // (depth taps: per-lane source planes are taken relative to b_base)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.weight, 0, clamp_bytes((int64_t)a.ksize * a.ksize * a.cout_s * p.cin_s * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(a.src0) + b_base * s0_elems * ES), 0, clamp_bytes((a.batch - b_base) * s0_elems * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? reinterpret_cast<const char*>(a.src1) + b_base * s1_elems * ES : reinterpret_cast<const char*>(a.src0)), 0,
      a.src1 ? clamp_bytes((a.batch - b_base) * s1_elems * ES) : 0u, 0x00020000);

  // loaders.  Weights: thread -> (16-byte chunk wcc = 8 k-values, rows wr0 + 32 i).  Activations: fp32 in memory: thread ->
  // (16-byte chunk acc4 = 4 k-values, rows ar0 + 16 i); operand type in memory (SRCH): (chunk = 8 k-values, rows ar0 + 32 i).
  const int wcc = tid & 7, wr0 = tid >> 3;
  const int acc4 = SRCH ? (tid & 7) : (tid & 15), ar0 = SRCH ? (tid >> 3) : (tid >> 4);
  constexpr int ARS = SRCH ? 32 : 16;      // row step between a thread's activation loads
  unsigned voffW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = m0 + wr0 + 32 * i;
    voffW[i] = co < a.cout_s ? (unsigned)((co * p.cin_s + wcc * 8) * 2) : OOB;
  }
  int prel[NA], ihb[NA], iwb[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int n = n0 + ar0 + ARS * i;
    const bool pv = n < p.npix;
    const int nn = pv ? n : 0;
    const int b = nn / hw_out;
    const int rem = nn - b * hw_out;
    const int oh = rem / a.wout;
    bool dok;
    const int ps = az_depth_plane(a, b, dok) - b_base;
    prel[i] = pv && dok ? ps : -1;  // (source plane relative to the descriptor's first one; -1: reads zeros)
    ihb[i] = oh * a.stride - a.pad;
    iwb[i] = (rem - oh * a.wout) * (a.aniso ? a.stride_w : a.stride) - a.pad;
  }

  const int nk_tap = p.nkc0 + p.nkc1;
  int it_tap = kt_begin / nk_tap;
  int it_r = kt_begin - it_tap * nk_tap;
  int it_src = it_r >= p.nkc0 ? 1 : 0;
  int it_kc = it_src ? it_r - p.nkc0 : it_r;
  unsigned voffA[NA];
  auto set_tap_src = [&]() __attribute__((always_inline)) {
    const int ky = it_tap / a.ksize;
    const int kx = it_tap - ky * a.ksize;
    const int cs = it_src ? a.c1s : a.c0s;
    const int up = it_src ? a.up1 : a.up0;
    const int upw = a.aniso ? (it_src ? a.up1_w : a.up0_w) : up;
    const int hs = it_src ? a.h1 : a.h0;
    const int ws = it_src ? a.w1 : a.w0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int ih = wrap_coord(ihb[i] + ky, a.hin, a.pad_mode);
      const int iw = wrap_coord(iwb[i] + kx, a.win, a.pad_mode);
      const bool ok = prel[i] >= 0 && (unsigned)ih < (unsigned)a.hin && (unsigned)iw < (unsigned)a.win;
      const int pix = (prel[i] * hs + (ih >> up)) * ws + (iw >> upw);
      voffA[i] = ok ? (unsigned)((pix * cs + acc4 * ACH) * ES) : OOB;
    }
  };

  float4 ra[4];   // 8 two-byte weights each
  float4 rb[NA];  // 4 fp32 activations each (SRCH: 8 two-byte ones)
  auto load_tile = [&]() __attribute__((always_inline)) {
    const int cs = it_src ? a.c1s : a.c0s;
    const int kbase = it_kc * HBK;
    const int kglob = (it_src ? a.c0s : 0) + kbase;
    const unsigned soffW = (unsigned)(((int64_t)it_tap * a.cout_s * p.cin_s + kglob) * 2);
    const unsigned soffA = (unsigned)(kbase * ES);
    const bool kv = kbase + acc4 * ACH < cs;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = buf_ld4(rw, voffW[i], soffW);
    if (it_src) {
#pragma unroll
      for (int i = 0; i < NA; ++i) rb[i] = buf_ld4(rs1, kv ? voffA[i] : OOB, soffA);
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) rb[i] = buf_ld4(rs0, kv ? voffA[i] : OOB, soffA);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++it_kc;
    if (it_kc == (it_src ? p.nkc1 : p.nkc0)) {
      it_kc = 0;
      ++it_src;
      if (it_src == 2 || p.nkc1 == 0) {
        it_src = 0;
        ++it_tap;
      }
      set_tap_src();
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    unsigned short* As = hsm + buf * HTILE;
    unsigned short* Bs = As + BM * HLDS;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(As + (wr0 + 32 * i) * HLDS + wcc * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if constexpr (SRCH) {
        *reinterpret_cast<float4*>(Bs + (ar0 + ARS * i) * HLDS + acc4 * 8) = rb[i];
        continue;
      }
      const f32x4v v = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
      if constexpr (F16) {
        const f16x4 hv = __builtin_convertvector(v, f16x4);
        *reinterpret_cast<f16x4*>(Bs + (ar0 + ARS * i) * HLDS + acc4 * 4) = hv;
      } else {
        const bf16x4 hv = __builtin_convertvector(v, bf16x4);
        *reinterpret_cast<bf16x4*>(Bs + (ar0 + ARS * i) * HLDS + acc4 * 4) = hv;
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment: row = lane & 31, the lane half (lane >> 5) picks which 8 of the 16 k-values of a K-step
  const int frag_off = (lane & 31) * HLDS + (lane >> 5) * 8;

  if (kt_begin < kt_end) {
    set_tap_src();
    load_tile();
    store_tile(0);
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) {
      advance();
      load_tile();
    }
    const unsigned short* As = hsm + buf * HTILE + (wc * 64) * HLDS + frag_off;
    const unsigned short* Bs = hsm + buf * HTILE + BM * HLDS + (wp * 64) * HLDS + frag_off;
#pragma unroll
    for (int ks = 0; ks < HBK / 16; ++ks) {
      if constexpr (F16) {
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(As + ks * 16);
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(As + 32 * HLDS + ks * 16);
        const f16x8 b0 = *reinterpret_cast<const f16x8*>(Bs + ks * 16);
        const f16x8 b1 = *reinterpret_cast<const f16x8*>(Bs + 32 * HLDS + ks * 16);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
      } else {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(As + ks * 16);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(As + 32 * HLDS + ks * 16);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bs + ks * 16);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bs + 32 * HLDS + ks * 16);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (a.dst_dtype) store_acc_tiles<F16 ? 2 : 1>(p, acc, m0, n0, wc, wp, lane, reinterpret_cast<float*>(hsm));
  else store_acc_tiles(p, acc, m0, n0, wc, wp, lane, reinterpret_cast<float*>(hsm));
}

// =================================================================================================
// fp32 operands on the bf16 matrix pipe ("bf16x3"): every fp32 value is split EXACTLY into three bf16 pieces
// (x = x1 + x2 + x3: three 8-bit slices of the 24-bit significand, by truncation), and a product is evaluated as the six
// largest of the nine partial products,  x1 y1 + (x1 y2 + x2 y1) + (x1 y3 + x2 y2 + x3 y1),  each an exact 16-bit
// product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are <= 3 * 2^-24 |x y|, i.e. at the level
// of ONE fp32 rounding of the product; measured against fp64 the result is as accurate as the fp32 MFMA kernel and
// more accurate than the Winograd form (tests/test_gpu_kernels.py::test_conv2d_x3_accuracy).  Six bf16 MFMAs per 16
// channels take 6 x 32 cycles against 8 x 64 for v_mfma_f32_32x32x2_f32: 0.375 x the matrix-pipe time.
// Weights are split once (az_pack_conv_weight_x3_f32: [piece][tap][cout_s][cin_s] bf16); activations stay fp32 in HBM
// and are split by the loader threads while they are staged (2 ANDs, 2 subtractions, 1.5 byte-permutes per value).
// Tile 128 couts x 128 pixels, K tile = 32 channels of one tap; LDS: 6 planes of 128 rows x 64 B = 48 KB, the four
// 16-byte chunks of a row XOR-swizzled with bits 2..3 of the row so that both the loader's ds_write_b128 (4 rows x 4
// chunks per 16 lanes) and the fragment ds_read_b128 (16 rows x 1 chunk) are bank-conflict-free (the padded 80-byte
// rows of the first version had 2-way write conflicts: SQ_LDS_BANK_CONFLICT = 11 % of the kernel time);
// single-buffered with register prefetch, two workgroups per CU.
constexpr int XBK = 32;                       // k tile
constexpr int XLDS = 32;                      // LDS row = the 32 k-values of a tile (64 B), 16-byte chunks XOR-swizzled by row
constexpr int XPLANE = 128 * XLDS;            // 2-byte elements per (operand, piece) plane
constexpr int X_LDS_BYTES = (BN * OSTR + BN) * 4;  // the epilogue's exchange buffer (68,096 B) >= 6 planes (61,440 B)
static_assert(6 * XPLANE * 2 <= X_LDS_BYTES, "stage planes must fit under the epilogue buffer");

__device__ __forceinline__ void split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  az_split3(x0, x1, p1, p2, p3);  // (common.h: shared with the attention kernel)
}

// One partial product on the pipe of the mode: bf16 pieces (bf16x3) or IEEE half pieces (f16x2); the operands are 16-byte fragments.
template <bool H2>
__device__ __forceinline__ f32x16 x3_mfma(const uint4& fa, const uint4& fb, const f32x16& c) {
  if constexpr (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), c, 0, 0, 0);
}
// The partial products of a mode as (weight piece, activation piece) pairs, smallest terms first.
//   bf16x3: the six largest of the nine products of 3 x 3 exact pieces.
//   f16x2 (H2): weights [wh | wl | wh / 2^11], activations [h | l = (x' - h) 2^11]:  wl h + (wh / 2^11) l + wh h.
template <bool H2> struct X3Prod;
template <> struct X3Prod<false> { static constexpr int N = 6; static constexpr int PA[6] = {2, 1, 0, 1, 0, 0}; static constexpr int PB[6] = {0, 1, 2, 0, 1, 0}; };
template <> struct X3Prod<true> { static constexpr int N = 3; static constexpr int PA[6] = {1, 2, 0, 0, 0, 0}; static constexpr int PB[6] = {0, 1, 0, 0, 0, 0}; };

// H2: the "f16x2" form (az_conv2d_f16x2_f32) -- the same tile, stages and epilogue; two activation planes instead of three, three
// matrix instructions per 16 channels instead of six, the accumulators scaled by p.out_scale behind the K loop.
template <bool H2>
__global__ __launch_bounds__(256, 2) void conv_igemm_x3_kernel(ConvP p) {
  __shared__ __attribute__((aligned(16))) float xsmf[X_LDS_BYTES / 4];
  unsigned short* xsm = reinterpret_cast<unsigned short*>(xsmf);
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wc = wid >> 1;
  const int wp = wid & 1;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tile_n = wg / p.tiles_m;
  const int tile_m = wg - tile_n * p.tiles_m;
  const int m0 = (p.m_tile0 + tile_m) * BM;
  const int n0 = tile_n * BN;
  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);
  // H2: the activation scale of this launch (fixed, or from the sources' absmax slots: common.h)
  const float pin = H2 ? az_f16x2_in_scale(a.in_absmax0, a.in_absmax1, 1.f, lane) : 1.f;
  const float pin2k = pin * 2048.f;

  const int hw_out = a.hout * a.wout;
  const int b_first = n0 / hw_out;
  const int64_t s0_elems = (int64_t)a.h0 * a.w0 * a.c0s;
  const int64_t s1_elems = (int64_t)a.h1 * a.w1 * a.c1s;
  const int64_t wplane = (int64_t)a.ksize * a.ksize * a.cout_s * p.cin_s;  // elements per weight piece
  auto clamp_bytes = [](int64_t e) { return (unsigned)(e <= 0 ? 0 : (e > AZ_RSRC_CLAMP ? AZ_RSRC_CLAMP : e)); };
  const int b_base = az_depth_base(a, b_first);  // (depth taps: per-lane source planes are taken relative to b_base)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, clamp_bytes(3 * wplane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src0 + b_base * s0_elems), 0, clamp_bytes((a.batch - b_base) * s0_elems * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 + b_base * s1_elems : a.src0), 0,
      a.src1 ? clamp_bytes((a.batch - b_base) * s1_elems * 4) : 0u, 0x00020000);

  // loaders: thread -> (8 consecutive k-values kc8, rows r0 + 64 i) of both operands
  const int kc8 = tid & 3, r0 = tid >> 2;
  unsigned voffW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int co = m0 + r0 + 64 * i;
    voffW[i] = co < a.cout_s ? (unsigned)((co * p.cin_s + kc8 * 8) * 2) : OOB;
  }
  int prel[2], ihb[2], iwb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + r0 + 64 * i;
    const bool pv = n < p.npix;
    const int nn = pv ? n : 0;
    const int b = nn / hw_out;
    const int rem = nn - b * hw_out;
    const int oh = rem / a.wout;
    bool dok;
    const int ps = az_depth_plane(a, b, dok) - b_base;
    prel[i] = pv && dok ? ps : -1;  // (source plane relative to the descriptor's first one; -1: reads zeros)
    ihb[i] = oh * a.stride - a.pad;
    iwb[i] = (rem - oh * a.wout) * (a.aniso ? a.stride_w : a.stride) - a.pad;
  }

  const int nk_tap = p.nkc0 + p.nkc1;
  int it_tap = kt_begin / nk_tap;
  int it_r = kt_begin - it_tap * nk_tap;
  int it_src = it_r >= p.nkc0 ? 1 : 0;
  int it_kc = it_src ? it_r - p.nkc0 : it_r;
  unsigned voffA[2];
  auto set_tap_src = [&]() __attribute__((always_inline)) {
    const int ky = it_tap / a.ksize;
    const int kx = it_tap - ky * a.ksize;
    const int cs = it_src ? a.c1s : a.c0s;
    const int up = it_src ? a.up1 : a.up0;
    const int upw = a.aniso ? (it_src ? a.up1_w : a.up0_w) : up;
    const int hs = it_src ? a.h1 : a.h0;
    const int ws = it_src ? a.w1 : a.w0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ih = wrap_coord(ihb[i] + ky, a.hin, a.pad_mode);
      const int iw = wrap_coord(iwb[i] + kx, a.win, a.pad_mode);
      const bool ok = prel[i] >= 0 && (unsigned)ih < (unsigned)a.hin && (unsigned)iw < (unsigned)a.win;
      const int pix = (prel[i] * hs + (ih >> up)) * ws + (iw >> upw);
      voffA[i] = ok ? (unsigned)((pix * cs + kc8 * 8) * 4) : OOB;
    }
  };

  float4 ra[3][2];  // [piece][row]: 8 two-byte weights each
  float4 rb[2][2];  // [row][half]: 4 fp32 activations each
  auto load_tile = [&]() __attribute__((always_inline)) {
    const int cs = it_src ? a.c1s : a.c0s;
    const int kbase = it_kc * XBK;
    const int kglob = (it_src ? a.c0s : 0) + kbase;
    const int64_t wtap = (int64_t)it_tap * a.cout_s * p.cin_s + kglob;
    const unsigned soffA = (unsigned)(kbase * 4);
    const bool kv0 = kbase + kc8 * 8 < cs, kv1 = kbase + kc8 * 8 + 4 < cs;
#pragma unroll
    for (int pl = 0; pl < (H2 ? 2 : 3); ++pl) {  // (H2: the third plane, wh / 2^11, is derived from the first in store_tile)
      const unsigned soffW = (unsigned)((pl * wplane + wtap) * 2);
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[pl][i] = buf_ld4(rw, voffW[i], soffW);
    }
    if (it_src) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rb[i][0] = buf_ld4(rs1, kv0 ? voffA[i] : OOB, soffA);
        rb[i][1] = buf_ld4(rs1, kv1 ? voffA[i] + 16u : OOB, soffA);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rb[i][0] = buf_ld4(rs0, kv0 ? voffA[i] : OOB, soffA);
        rb[i][1] = buf_ld4(rs0, kv1 ? voffA[i] + 16u : OOB, soffA);
      }
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++it_kc;
    if (it_kc == (it_src ? p.nkc1 : p.nkc0)) {
      it_kc = 0;
      ++it_src;
      if (it_src == 2 || p.nkc1 == 0) {
        it_src = 0;
        ++it_tap;
      }
      set_tap_src();
    }
  };
  const int wsw = (kc8 ^ ((r0 >> 2) & 3)) * 8;  // swizzled chunk of this thread's rows (r0 and r0 + 64 share bits 2..3)
  auto store_tile = [&]() __attribute__((always_inline)) {
    if constexpr (H2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[2][i] = __builtin_bit_cast(float4, __builtin_bit_cast(f16x8, ra[0][i]) * (_Float16)0.00048828125f);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<float4*>(xsm + pl * XPLANE + (r0 + 64 * i) * XLDS + wsw) = ra[pl][i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float x[8] = {rb[i][0].x, rb[i][0].y, rb[i][0].z, rb[i][0].w, rb[i][1].x, rb[i][1].y, rb[i][1].z, rb[i][1].w};
      unsigned q[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (H2) az_split2h(x[2 * j], x[2 * j + 1], pin, pin2k, q[0][j], q[1][j]);
        else split3(x[2 * j], x[2 * j + 1], q[0][j], q[1][j], q[2][j]);
      }
#pragma unroll
      for (int pl = 0; pl < (H2 ? 2 : 3); ++pl)
        *reinterpret_cast<uint4*>(xsm + (3 + pl) * XPLANE + (r0 + 64 * i) * XLDS + wsw) =
            make_uint4(q[pl][0], q[pl][1], q[pl][2], q[pl][3]);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = (lane & 31) * XLDS;
  const int fsw = ((lane & 31) >> 2) & 3;  // the row's swizzle (tile bases are multiples of 32 rows)
  const int fch[2] = {((0 + (lane >> 5)) ^ fsw) * 8, ((2 + (lane >> 5)) ^ fsw) * 8};  // K step ks: chunk 2 ks + lane half
  const unsigned short* As = xsm + (wc * 64) * XLDS + frag_row;               // + piece * XPLANE + tile * 32 * XLDS + chunk
  const unsigned short* Bs = xsm + 3 * XPLANE + (wp * 64) * XLDS + frag_row;

  if (kt_begin < kt_end) {
    set_tap_src();
    load_tile();
    store_tile();
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) {
      advance();
      load_tile();  // in flight under the 48 MFMAs below
    }
#pragma unroll
    for (int ks = 0; ks < XBK / 16; ++ks) {
      uint4 fa[3][2], fb[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[pl][t] = *reinterpret_cast<const uint4*>(As + pl * XPLANE + t * 32 * XLDS + fch[ks]);
          if (!H2 || pl < 2) fb[pl][t] = *reinterpret_cast<const uint4*>(Bs + pl * XPLANE + t * 32 * XLDS + fch[ks]);
        }
      // smallest partial products first
      using PR = X3Prod<H2>;
#pragma unroll
      for (int t = 0; t < PR::N; ++t) {
        acc[0][0] = x3_mfma<H2>(fa[PR::PA[t]][0], fb[PR::PB[t]][0], acc[0][0]);
        acc[0][1] = x3_mfma<H2>(fa[PR::PA[t]][0], fb[PR::PB[t]][1], acc[0][1]);
        acc[1][0] = x3_mfma<H2>(fa[PR::PA[t]][1], fb[PR::PB[t]][0], acc[1][0]);
        acc[1][1] = x3_mfma<H2>(fa[PR::PA[t]][1], fb[PR::PB[t]][1], acc[1][1]);
      }
    }
    __syncthreads();  // every wave has read this K tile
    if (more) store_tile();
    __syncthreads();
  }
  if constexpr (H2) {
    const float osc = __fdiv_rn(p.out_scale, pin);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = acc[i][j] * osc;
  }
  store_acc_tiles(p, acc, m0, n0, wc, wp, lane, xsmf);
}

// =================================================================================================
// bf16x3 token GEMM / 1x1 convolution on a 256 x 256 tile (conv_gemm_x3_big_kernel).  The 128 x 128 kernel above is power-bound
// with about half of its energy outside the matrix pipe (profiles/r04_x3_gemm_ablation.txt): per MFMA it moves 0.5 fragment
// reads, and per 128 x 128 x 16 unit 20 KB of operand fill, 24 KB of LDS stores and one activation split.  Here 8 waves share
// one 256-cout x 256-pixel tile (wave = 128 couts x 64 pixels = 4 x 2 accumulators): 0.375 fragment reads per MFMA, and per
// unit 10 KB of fill, 12 KB of LDS stores, half a split; K step = 16 channels, two LDS stages of
// [operand][piece][256 rows][32 B] (96 KB), ONE barrier per 48 MFMAs, the next step's operands split and stored into the other
// stage under the MFMAs, the one after that in flight from L2.  Rows are 32 bytes; the two 16-byte chunks of a row are
// swapped in rows with bit 3 set (ds_read_b128 lane groups {0-3, 12-15, 20-27}, ... then cover all 64 banks).
// One workgroup of 512 threads per CU.  1 x 1, stride 1, one source, c0s % 16 == 0, NHWC destination.
constexpr int GB = 256;                      // tile edge (couts and pixels)
constexpr int GBK = 16;                      // K step
constexpr int GPLANE = GB * GBK * 2;         // bytes per (operand, piece) plane of a stage: 8 KB
constexpr int GSTAGE = 6 * GPLANE;           // 48 KB
constexpr int GOSTR = 128 + 4;               // floats per pixel row of the epilogue's exchange buffer (one 128-cout pass)
constexpr int G_LDS_BYTES = (GB * GOSTR + GB) * 4;  // 136,192 B >= 2 stages (98,304 B)
static_assert(2 * GSTAGE <= G_LDS_BYTES, "stages must fit under the epilogue buffer");

// Epilogue of a 256 x 256 tile held as 8 waves x (128 couts x 64 pixels): two passes of 128 couts through the exchange buffer
// [pixel][128 + 4] (the layout of store_acc_tiles), all 512 threads storing (bias / activation / gate / residual / SwiGLU / q-k
// preparation / split-K slabs: epilogue_store_batch).
// NCT = 32-cout MFMA tiles per wave: 4 (256-cout tile, passes of 128) or 3 (192-cout tile, passes of 96 couts = 24 quads per pixel
// row, 21 rows per sweep of the 512 threads, 8 of them idle).
template <int NCT, int IO = 0>
__device__ __forceinline__ void gemm_big_epilogue(const ConvP& p, const f32x16 (&acc)[NCT][2], int m0, int n0, int wc, int wp, int lane,
                                                  int tid, float* gsmf) {
  constexpr int PW = 32 * NCT;         // couts per pass
  constexpr int Q = PW / 4;            // channel quads per pixel row
  constexpr int RPI = 512 / Q;         // pixel rows per sweep
  constexpr int OST = PW + 4;          // floats per pixel row of the exchange buffer
  const AzConvArgs& a = p.a;
  int* pimg = reinterpret_cast<int*>(gsmf + GB * GOSTR);  // image index of each pixel of the tile (-1: past the end)
  if (tid < GB) {
    const int n = n0 + tid;
    pimg[tid] = n < p.npix ? n / (a.hout * a.wout) : -1;
  }
#pragma unroll 1
  for (int cb = 0; cb < 2; ++cb) {
    if (wc == cb) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        float* orow = gsmf + (wp * 64 + pt * 32 + (lane & 31)) * OST + 4 * (lane >> 5);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(orow + ct * 32 + 8 * q) =
                make_float4(acc[ct][pt][4 * q + 0], acc[ct][pt][4 * q + 1], acc[ct][pt][4 * q + 2], acc[ct][pt][4 * q + 3]);
      }
    }
    __syncthreads();
    const int cq = tid % Q;  // the same channel quad in every iteration
    const int pr = tid / Q;  // pixel row inside a sweep (>= RPI: an idle thread of the 24-quad form)
    const int co = m0 + cb * PW + cq * 4;
    if (co < a.cout_s && pr < RPI) {
      constexpr int NIT = (GB + RPI - 1) / RPI, NB = 4;  // (8 per batch spills here: the other cout half's accumulator registers are live)
#pragma unroll 1
      for (int it0 = 0; it0 < NIT; it0 += NB) {
        int n[NB], b[NB];
        float4 v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int px = (it0 + i) * RPI + pr;
          const bool in = px < GB;
          b[i] = in ? pimg[px] : -1;
          n[i] = b[i] >= 0 ? n0 + px : -1;
          v[i] = in ? *reinterpret_cast<const float4*>(gsmf + px * OST + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        epilogue_store_batch<NB, false, IO>(a, n, b, co, v, (int64_t)blockIdx.y * p.npix);
      }
    }
    __syncthreads();
  }
}


// NCT = 32-cout MFMA tiles per wave: 4 = the 256-cout tile, 3 = a 192-cout tile (768 = 4 x 192: whole rounds where 3 x 256 leaves a
// quarter of the CUs idle).  TAPS: k x k filters with a stride and zero padding (the strided 3x3 convolutions of a UNet's
// descent): the K walk is tap-major, the loader thread's pixel row is fixed, so a tap is one offset and one bounds test per step.
// H2: the "f16x2" form (az_conv2d_f16x2_f32): two activation planes, 24 instead of 48 matrix instructions per step (NCT = 4).
template <int NCT, bool TAPS = false, bool H2 = false>
__global__ __launch_bounds__(512, 1) void conv_gemm_x3_big_kernel(ConvP p) {
  constexpr int CT = 64 * NCT;  // couts per tile
  __shared__ __attribute__((aligned(16))) float gsmf[G_LDS_BYTES / 4];
  char* const smem = reinterpret_cast<char*>(gsmf);
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wc = wid >> 2;  // cout half (128)
  const int wp = wid & 3;   // pixel quarter (64)

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tile_n = wg / p.tiles_m;
  const int tile_m = wg - tile_n * p.tiles_m;
  const int m0 = tile_m * CT;
  const int n0 = tile_n * GB;
  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);
  const int nk = kt_end - kt_begin;
  // H2: the activation scale of this launch (fixed, or from the sources' absmax slots: common.h)
  const float pin = H2 ? az_f16x2_in_scale(a.in_absmax0, a.in_absmax1, 1.f, lane) : 1.f;
  const float pin2k = pin * 2048.f;

  const int64_t wtap = (int64_t)a.cout_s * p.cin_s;               // elements per (piece, tap)
  const int64_t wplane = (int64_t)a.ksize * a.ksize * wtap;       // elements per weight piece
  const int64_t spix = (int64_t)a.batch * a.h0 * a.w0;             // source pixels (= p.npix for a 1x1, stride-1 launch)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, (unsigned)(3 * wplane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (unsigned)(spix * a.c0s * 4), 0x00020000);
  // (two sources -- a channel concatenation read in place, the 1x1 skip convolutions of ADM's decoder: K steps [0, nkc0) walk
  //  source 0, the rest source 1; the packed weights hold source 1's channels from column c0s on)
  const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 : a.src0), 0, a.src1 ? (unsigned)(spix * a.c1s * 4) : 0u, 0x00020000);

  // loaders: thread -> row tid >> 1 of both operands; weights: chunk tid & 1 (8 k-values) of each piece, activations: the 8
  // fp32 channels of that chunk (two adjacent lanes read one 64-byte unit)
  const int lrow = tid >> 1, lch = tid & 1;
  // (TAPS) the output pixel of this thread's row: image plane base, top-left input coordinate of its window
  const bool rowok = n0 + lrow < p.npix;
  const int hw_out = a.hout * a.wout;
  const int nn_l = rowok ? n0 + lrow : 0;
  const int b_l = nn_l / hw_out, rem_l = nn_l - b_l * hw_out, oh_l = rem_l / a.wout;
  const int ihb = oh_l * a.stride - a.pad, iwb = (rem_l - oh_l * a.wout) * a.stride - a.pad, pb_l = b_l * a.h0;
  const int nk_tap = p.nkc0 + p.nkc1;
  const int co_l = min(m0 + lrow, a.cout_s - 1);  // (rows past the edge: a valid duplicate, never stored)
  const int px_l = min(n0 + lrow, p.npix - 1);
  const unsigned voffW = lrow < CT ? (unsigned)(((int64_t)co_l * p.cin_s + lch * 8) * 2) : OOB;  // (a 192-cout tile: rows 192 .. 255 stage zeros)
  const unsigned voffX = (unsigned)(((int64_t)px_l * a.c0s + lch * 8) * 4);
  const unsigned voffX1 = (unsigned)(((int64_t)px_l * a.c1s + lch * 8) * 4);
  const int lds_row = lrow * 32 + ((lch ^ ((lrow >> 3) & 1)) * 16);  // byte offset inside a plane

  // H2: the third weight plane (wh / 2^11) is the first with another exponent: derived by the loader thread (four v_pk_mul_f16, the
  // packing's rounding bit for bit) instead of loaded -- 4 instead of 5 16-byte loads per thread and step
  constexpr int NWL = H2 ? 2 : 3;  // weight planes loaded
  float4 rwt[3], rxa[2];
  auto load_step = [&](int kt) __attribute__((always_inline)) {
    if constexpr (TAPS) {
      const int tap = kt / nk_tap, kr = kt - tap * nk_tap;  // (wave-uniform)
      const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
#pragma unroll
      for (int pl = 0; pl < NWL; ++pl) rwt[pl] = buf_ld4(rw, voffW, (unsigned)((pl * wplane + tap * wtap + (int64_t)kr * GBK) * 2));
      const int ih = ihb + ky, iw = iwb + kx;
      const bool ok = rowok && (unsigned)ih < (unsigned)a.h0 && (unsigned)iw < (unsigned)a.w0;  // (zero padding: reads zeros)
      const int pix = (pb_l + ih) * a.w0 + iw;
      const bool s1 = kr >= p.nkc0;
      const __amdgpu_buffer_rsrc_t r = s1 ? rx1 : rx;
      const unsigned vo = ok ? (unsigned)(((int64_t)pix * (s1 ? a.c1s : a.c0s) + lch * 8) * 4) : OOB;
      const unsigned so = (unsigned)((s1 ? kr - p.nkc0 : kr) * GBK * 4);
      rxa[0] = buf_ld4(r, vo, so);
      rxa[1] = buf_ld4(r, ok ? vo + 16u : OOB, so);
    } else {
#pragma unroll
      for (int pl = 0; pl < NWL; ++pl) rwt[pl] = buf_ld4(rw, voffW, (unsigned)((pl * wplane + (int64_t)kt * GBK) * 2));
      const bool s1 = kt >= p.nkc0;  // (wave-uniform; selects, not a branch: the iteration stays one basic block)
      const __amdgpu_buffer_rsrc_t r = s1 ? rx1 : rx;
      const unsigned vo = s1 ? voffX1 : voffX, so = (unsigned)((s1 ? kt - p.nkc0 : kt) * GBK * 4);
      rxa[0] = buf_ld4(r, vo, so);
      rxa[1] = buf_ld4(r, vo + 16u, so);
    }
  };
  auto store_step = [&](int buf) __attribute__((always_inline)) {
    char* st = smem + buf * GSTAGE + lds_row;
    if constexpr (H2) rwt[2] = __builtin_bit_cast(float4, __builtin_bit_cast(f16x8, rwt[0]) * (_Float16)0.00048828125f);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<float4*>(st + pl * GPLANE) = rwt[pl];
    const float x[8] = {rxa[0].x, rxa[0].y, rxa[0].z, rxa[0].w, rxa[1].x, rxa[1].y, rxa[1].z, rxa[1].w};
    unsigned q[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (H2) az_split2h(x[2 * j], x[2 * j + 1], pin, pin2k, q[0][j], q[1][j]);
      else split3(x[2 * j], x[2 * j + 1], q[0][j], q[1][j], q[2][j]);
    }
#pragma unroll
    for (int pl = 0; pl < (H2 ? 2 : 3); ++pl)
      *reinterpret_cast<uint4*>(st + (3 + pl) * GPLANE) = make_uint4(q[pl][0], q[pl][1], q[pl][2], q[pl][3]);
  };

  f32x16 acc[NCT][2];
#pragma unroll
  for (int i = 0; i < NCT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row (lane & 31) of a 32-row MFMA tile, chunk lane >> 5 (swapped in rows with bit 3 set)
  const int frow = lane & 31;
  const int foff = frow * 32 + (((lane >> 5) ^ ((frow >> 3) & 1)) * 16);
  const char* As = smem + (wc * (CT / 2)) * 32 + foff;               // + buf * GSTAGE + piece * GPLANE + tile * 1024
  const char* Bs = smem + 3 * GPLANE + (wp * 64) * 32 + foff;

  if (nk > 0) {
    load_step(kt_begin);
    store_step(0);
    load_step(min(kt_begin + 1, kt_end - 1));
  }
  __syncthreads();
#pragma unroll 1
  for (int i = 0; i < nk; ++i) {
    const int buf = i & 1;
    constexpr int NPB = H2 ? 2 : 3;  // activation planes
    uint4 fa[3][NCT], fb[NPB][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int t = 0; t < NCT; ++t) fa[pl][t] = *reinterpret_cast<const uint4*>(As + buf * GSTAGE + pl * GPLANE + t * 1024);
      if (pl < NPB) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[pl][t] = *reinterpret_cast<const uint4*>(Bs + buf * GSTAGE + pl * GPLANE + t * 1024);
      }
    }
    // (unconditional, so that the iteration is ONE basic block the scheduler can interleave: the last two iterations restage
    // the final step into a stage nobody reads)
    store_step(buf ^ 1);                               // step i + 1 (in registers since the previous iteration) -> the other stage
    load_step(min(kt_begin + i + 2, kt_end - 1));      // in flight under the MFMAs below and the next iteration's first ones
    using PR = X3Prod<H2>;  // smallest partial products first
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int ci = 0; ci < NCT; ++ci)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj)
          acc[ci][pj] = x3_mfma<H2>(fa[PR::PA[t]][ci], fb[PR::PB[t]][pj], acc[ci][pj]);
    // Issue order: the 18 fragment reads, then the staging of the next step (44 split instructions, 6 LDS stores, 5 loads) spread
    // under the 48 MFMAs -- the matrix pipe takes 32 cycles per instruction, ~5 other issues fit in each gap -- instead of in
    // front of them (the compiler's own order: the pipe idles while both waves of a SIMD stage).
    constexpr int NM = PR::N * 2 * NCT;  // MFMAs of the step
    if constexpr (!H2) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (NCT + 2), 0);  // DS reads
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
        if (k < NM - 8) __builtin_amdgcn_sched_group_barrier(0x002, NCT == 4 ? 2 : 3, 0);  // two (three) vector instructions
        if (k >= 8 && k < NM - 8 && (k & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS store
        if (k >= NM - 8 && k < NM - 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // a buffer load
      }
    } else {
      // half as many matrix instructions for the same staging: the three weight planes' stores (no arithmetic in front of them)
      // early, the split (28 vector instructions) three to a gap, the two activation stores behind it, the 4 loads last
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * NCT + 4, 0);  // DS reads
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
        if (k < NM - 6) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three vector instructions
        if (k == 1 || k == 3 || k == 5 || k == NM - 9 || k == NM - 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS store
        if (k >= NM - 5 && k < NM - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // a buffer load (four: the third weight plane is derived)
      }
    }
    // (measured alternatives: the staging ten MFMAs later 388 vs 371 us, one vector instruction per gap all along 380, on 16384 x 768 -> 3072;
    //  the compiler's own order, tools/ablate.py x3big_nosched: 411)
    __syncthreads();  // every wave has read stage `buf`; the other stage is complete
  }

  if constexpr (H2) {
    const float osc = __fdiv_rn(p.out_scale, pin);
#pragma unroll
    for (int i = 0; i < NCT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = acc[i][j] * osc;
  }
  gemm_big_epilogue<NCT>(p, acc, m0, n0, wc, wp, lane, tid, gsmf);
}

// =================================================================================================
// The 256 x 256 tile for modules cast to half precision (az_conv2d_{bf16,f16}_f32 on token GEMMs / 1x1 convolutions): the same
// 8 waves x (128 couts x 64 pixels), K steps of 64 channels (32 MFMAs per wave and barrier), two LDS stages of
// [operand][256 rows][128 B] (128 KB); weights arrive in the operand type, activations fp32 and are rounded to it (nearest even,
// as conv_igemm_half_kernel) on their way into LDS, under the MFMAs.  A row's eight 16-byte chunks are XOR-ed with bits 1..3 of
// the row: the 16 lanes of a ds_read_b128 group then hit 16 different slots of the 256-byte bank line.
constexpr int HGK = 64;                       // K step
constexpr int HGPLANE = GB * HGK * 2;         // bytes per operand plane of a stage: 32 KB
constexpr int HGSTAGE = 2 * HGPLANE;          // 64 KB
static_assert(2 * HGSTAGE <= G_LDS_BYTES, "stages must fit under the epilogue buffer");

// SRCH: the sources hold the operand type already (AzConvArgs.src_dtype = 1): a row's 64 k-values are 128 contiguous bytes, one
// 16-byte load per row and thread instead of two, stored to LDS as loaded.
// Measured and not kept in round 6 (profiles/r06_half_gemm.txt; all parity-green):
//  * the same tile as FOUR waves of 128 couts x 128 pixels (8 KB of fragment reads per 16 matrix instructions instead of 6 KB per 8):
//    10 - 27 % slower on every shape -- one wave per SIMD has nobody to cover its barrier and LDS round trips;
//  * two register sets for the typed operand loads (step i + 3 loaded during iteration i) and the compiler's own issue order instead
//    of the sched_group_barrier pattern: each looked 3 - 9 % ahead in isolated launches on one box, and same-box A/B of the
//    captured C3 step (234 - 235 against 237 - 239 images/s for this form) says they are not.
// Where its time is (timing ablations, same file): the load / stage / fragment-read / barrier skeleton alone is 60 - 66 % of a launch,
// the matrix instructions overlap it only partly (matrix pipe busy 0.42 on the active CUs), and with 768-channel K loops the
// epilogue -- a round's 32 MB of stores leaving in one burst -- is another 30 %.
template <bool F16, bool TAPS = false, bool SRCH = false>  // TAPS: k x k filters with a stride and zero padding, as in conv_gemm_x3_big_kernel
__global__ __launch_bounds__(512, 1) void conv_gemm_half_big_kernel(ConvP p) {
  constexpr int ES = SRCH ? 2 : 4;  // bytes per source element
  using H8 = typename std::conditional<F16, f16x8, bf16x8>::type;
  using H4 = typename std::conditional<F16, f16x4, bf16x4>::type;
  __shared__ __attribute__((aligned(16))) float gsmf[G_LDS_BYTES / 4];
  char* const smem = reinterpret_cast<char*>(gsmf);
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wc = wid >> 2;  // cout half (128)
  const int wp = wid & 3;   // pixel quarter (64)

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tile_n = wg / p.tiles_m;
  const int tile_m = wg - tile_n * p.tiles_m;
  const int m0 = tile_m * GB;
  const int n0 = tile_n * GB;
  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);
  const int nk = kt_end - kt_begin;

  const int64_t wtap = (int64_t)a.cout_s * p.cin_s;     // elements per tap
  const int64_t spix = (int64_t)a.batch * a.h0 * a.w0;   // source pixels (= p.npix for a 1x1, stride-1 launch)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, (unsigned)(a.ksize * a.ksize * wtap * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (unsigned)(spix * a.c0s * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 : a.src0), 0, a.src1 ? (unsigned)(spix * a.c1s * ES) : 0u, 0x00020000);

  // loaders: thread -> chunk tid & 7 (8 k-values) of rows (tid >> 3) + 64 i of both operands: the 8 lanes of a row read 128
  // (weights, half-precision activations) / 256 (fp32 activations) contiguous bytes
  const int lch = tid & 7, lr0 = tid >> 3;
  unsigned voffW[4], voffX[4], voffX1[4];
  int lds_row[4];
  int ihb[4], iwb[4], pb[4];  // (TAPS) top-left input coordinate of each row's window, its image's first source row; pb < 0: no pixel
  const int hw_out = a.hout * a.wout, nk_tap = p.nkc0 + p.nkc1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = lr0 + 64 * i;
    const int co = min(m0 + row, a.cout_s - 1);  // (rows past the edge: a valid duplicate, never stored)
    const int px = min(n0 + row, p.npix - 1);
    {
      const int b_ = px / hw_out, rem = px - b_ * hw_out, oh = rem / a.wout;
      ihb[i] = oh * a.stride - a.pad;
      iwb[i] = (rem - oh * a.wout) * a.stride - a.pad;
      pb[i] = n0 + row < p.npix ? b_ * a.h0 : -1;
    }
    voffW[i] = (unsigned)(((int64_t)co * p.cin_s + lch * 8) * 2);
    voffX[i] = (unsigned)(((int64_t)px * a.c0s + lch * 8) * ES);
    voffX1[i] = (unsigned)(((int64_t)px * a.c1s + lch * 8) * ES);
    lds_row[i] = row * 128 + ((lch ^ ((row >> 1) & 7)) * 16);
  }
  float4 rwt[4], rxa[4][2];
  auto load_step = [&](int kt) __attribute__((always_inline)) {
    if constexpr (TAPS) {
      const int tap = kt / nk_tap, kr = kt - tap * nk_tap;  // (wave-uniform)
      const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
      const bool s1 = kr >= p.nkc0;
      const __amdgpu_buffer_rsrc_t r = s1 ? rx1 : rx;
      const int cs = s1 ? a.c1s : a.c0s;
      const unsigned so = (unsigned)((s1 ? kr - p.nkc0 : kr) * HGK * ES);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rwt[i] = buf_ld4(rw, voffW[i], (unsigned)((tap * wtap + (int64_t)kr * HGK) * 2));
        const int ih = ihb[i] + ky, iw = iwb[i] + kx;
        const bool ok = pb[i] >= 0 && (unsigned)ih < (unsigned)a.h0 && (unsigned)iw < (unsigned)a.w0;  // (zero padding: reads zeros)
        const unsigned vo = ok ? (unsigned)((((int64_t)(pb[i] + ih) * a.w0 + iw) * cs + lch * 8) * ES) : OOB;
        rxa[i][0] = buf_ld4(r, vo, so);
        if constexpr (!SRCH) rxa[i][1] = buf_ld4(r, ok ? vo + 16u : OOB, so);
      }
    } else {
      const bool s1 = kt >= p.nkc0;  // (wave-uniform; selects, not a branch)
      const __amdgpu_buffer_rsrc_t r = s1 ? rx1 : rx;
      const unsigned so = (unsigned)((s1 ? kt - p.nkc0 : kt) * HGK * ES);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rwt[i] = buf_ld4(rw, voffW[i], (unsigned)(kt * HGK * 2));
        const unsigned vo = s1 ? voffX1[i] : voffX[i];
        rxa[i][0] = buf_ld4(r, vo, so);
        if constexpr (!SRCH) rxa[i][1] = buf_ld4(r, vo + 16u, so);
      }
    }
  };
  auto store_step = [&](int buf) __attribute__((always_inline)) {
    char* st = smem + buf * HGSTAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(st + lds_row[i]) = rwt[i];
      if constexpr (SRCH) {
        *reinterpret_cast<float4*>(st + HGPLANE + lds_row[i]) = rxa[i][0];
        continue;
      }
      const f32x4v lo = {rxa[i][0].x, rxa[i][0].y, rxa[i][0].z, rxa[i][0].w}, hi = {rxa[i][1].x, rxa[i][1].y, rxa[i][1].z, rxa[i][1].w};
      const H4 l4 = __builtin_convertvector(lo, H4), h4 = __builtin_convertvector(hi, H4);
      H8 v8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v8[j] = l4[j];
        v8[4 + j] = h4[j];
      }
      *reinterpret_cast<H8*>(st + HGPLANE + lds_row[i]) = v8;
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row (lane & 31) of a 32-row MFMA tile (tile bases are multiples of 32: the swizzle sees the row's own
  // bits 1..3), chunk 2 ks + (lane >> 5)
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const char* As = smem + (wc * 128 + frow) * 128;             // + buf * HGSTAGE + tile * 4096 + swizzled chunk * 16
  const char* Bs = smem + HGPLANE + (wp * 64 + frow) * 128;

  if (nk > 0) {
    load_step(kt_begin);
    store_step(0);
    load_step(min(kt_begin + 1, kt_end - 1));
  }
  __syncthreads();
#pragma unroll 1
  for (int i = 0; i < nk; ++i) {
    const int buf = i & 1;
    store_step(buf ^ 1);                               // step i + 1 (in registers since the previous iteration) -> the other stage
    load_step(min(kt_begin + i + 2, kt_end - 1));      // in flight under the MFMAs below and the next iteration's first ones
#pragma unroll
    for (int ks = 0; ks < HGK / 16; ++ks) {
      const int ch = ((2 * ks + (lane >> 5)) ^ fsw) * 16;
      H8 fa[4], fb[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const H8*>(As + buf * HGSTAGE + t * 4096 + ch);
#pragma unroll
      for (int t = 0; t < 2; ++t) fb[t] = *reinterpret_cast<const H8*>(Bs + buf * HGSTAGE + t * 4096 + ch);
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
          if constexpr (F16) acc[ci][pj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ci], fb[pj], acc[ci][pj], 0, 0, 0);
          else acc[ci][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ci], fb[pj], acc[ci][pj], 0, 0, 0);
        }
    }
    // issue order: the 24 fragment reads in four groups ahead of their MFMAs, the conversions and the 8 LDS stores of the next
    // step under the 32 MFMAs, its 12 loads last
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if ((k & 7) == 1 && k < 24) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // the next k-step's fragments
      if (k < 24) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                   // two vector instructions
      if (k >= 4 && k < 20 && (k & 1) == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS store
      if (k >= 20) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  // a buffer load
    }
    __syncthreads();  // every wave has read stage `buf`; the other stage is complete
  }
  if (a.dst_dtype) gemm_big_epilogue<4, F16 ? 2 : 1>(p, acc, m0, n0, wc, wp, lane, tid, gsmf);
  else gemm_big_epilogue<4>(p, acc, m0, n0, wc, wp, lane, tid, gsmf);
}

// =================================================================================================
// Narrow-output 3x3 convolution (cout_s == 4: the image head of a UNet, 256 -> 3 channels).  A 128-cout MFMA tile would
// waste 97 % of the matrix pipe on it and the Winograd form 94 %; with 27 x Cin multiply-adds per output pixel it is a
// VALU kernel: one thread per output pixel, the workgroup's 18 x 18 input halo staged through LDS in 16-channel chunks
// (pixel stride padded to 80 B -> conflict-free ds_read_b128), the weights (uniform
// across the wave) fetched by scalar loads and used as SGPR operands of the FMAs; the next chunk's global loads are in
// flight during the FMAs and 4 workgroups per CU overlap one another's barriers.  Same fused epilogue as the other kernels (the NCHW destination is coalesced here: lanes = pixels).
constexpr int HD_T = 16;               // output pixels per workgroup: 16 x 16
constexpr int HD_HALO = HD_T + 2;
constexpr int HD_KC = 16;              // input channels per chunk (4 workgroups per CU)
constexpr int HD_PS = HD_KC + 4;       // LDS pixel stride (floats)

template <int NCO>
__global__ __launch_bounds__(256) void conv_head_kernel(ConvP p, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float xs[HD_HALO * HD_HALO * HD_PS];  // 25,920 B
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int tx = tid & (HD_T - 1), ty = tid / HD_T;
  const int bid = blockIdx.x;
  const int txb = bid % tiles_x;
  const int rb = bid / tiles_x;
  const int tyb = rb % tiles_y;
  const int b = rb / tiles_y;
  const int oy0 = tyb * HD_T, ox0 = txb * HD_T;
  const float* __restrict__ w = a.weight;  // [tap][4][cin_s]
  const float* __restrict__ src = a.src0 + (int64_t)b * a.h0 * a.w0 * a.c0s;
  float acc[NCO];
#pragma unroll
  for (int co = 0; co < NCO; ++co) acc[co] = 0.f;

  // staging slots of this thread: element e = tid + 256 i of the [324 pixels][8 float4] chunk (chunk-invariant offsets)
  constexpr int NE = HD_HALO * HD_HALO * (HD_KC / 4);
  constexpr int NS = (NE + 255) / 256;
  int goff[NS], loff[NS];  // global float offset (-1: zero padding), LDS float offset (-1: no slot)
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int e = tid + 256 * i;
    const int pxi = e / (HD_KC / 4), q = e % (HD_KC / 4);
    const int hy = pxi / HD_HALO, hx = pxi - hy * HD_HALO;
    const int iy = wrap_coord(oy0 - 1 + hy, a.hin, a.pad_mode), ix = wrap_coord(ox0 - 1 + hx, a.win, a.pad_mode);
    const bool inb = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
    loff[i] = e < NE ? pxi * HD_PS + q * 4 : -1;
    goff[i] = (e < NE && inb) ? (iy * a.w0 + ix) * a.c0s + q * 4 : -1;
  }

  float4 v[NS];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
      v[i] = goff[i] >= 0 ? *reinterpret_cast<const float4*>(src + goff[i] + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  fetch(0);
  for (int kc = 0; kc < a.c0s; kc += HD_KC) {
    __syncthreads();  // the previous chunk has been consumed
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (loff[i] >= 0) *reinterpret_cast<float4*>(xs + loff[i]) = v[i];
    __syncthreads();
    if (kc + HD_KC < a.c0s) fetch(kc + HD_KC);  // in flight during this chunk's FMAs
    f32x2 part[NCO];  // per-chunk partial sums (2 x 72 terms, packed FMAs), added once: shorter rounding chains than one 9 Cin-long sum
#pragma unroll
    for (int co = 0; co < NCO; ++co) part[co] = f32x2{0.f, 0.f};
    // 9 groups (one per tap) of 16 channels: 48 weights in SGPRs (3 x s_load_dwordx16), 4 ds_read_b128, 48 FMAs.  The
    // group loop is NOT unrolled: hoisted scalar loads would only add lgkmcnt waits (SMEM and LDS share the counter)
    // -- the other waves of the SIMD cover the one scalar-cache round trip per group.
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = (tap * 11) >> 5;  // tap / 3 for tap < 9
      const float* xp = xs + ((ty + ky) * HD_HALO + tx + (tap - 3 * ky)) * HD_PS;
      const float* wt = w + (int64_t)tap * 4 * p.cin_s + kc;
      float wl[NCO][HD_KC];
#pragma unroll
      for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int j = 0; j < HD_KC; ++j) wl[co][j] = wt[co * p.cin_s + j];  // wave-uniform: scalar loads
#pragma unroll
      for (int q = 0; q < HD_KC / 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(xp + q * 4);
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
          part[co] = __builtin_elementwise_fma(f32x2{x.x, x.y}, f32x2{wl[co][q * 4 + 0], wl[co][q * 4 + 1]}, part[co]);
          part[co] = __builtin_elementwise_fma(f32x2{x.z, x.w}, f32x2{wl[co][q * 4 + 2], wl[co][q * 4 + 3]}, part[co]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[co] += part[co].x + part[co].y;
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= a.hout || ox >= a.wout) return;
  const int n = (b * a.hout + oy) * a.wout + ox;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int co = 0; co < NCO; ++co) o[co] = acc[co];
  epilogue_store_b(a, n, b, 0, make_float4(o[0], o[1], o[2], o[3]));
}

// =================================================================================================
// Image stem: 3x3, stride 1, pad 1 convolution of a PLANAR source with <= 4 channels (the latent itself: (B, C, H, W) as the
// sampler holds it) into the NHWC activation.  27 multiplies per output: no matrix work (the Winograd kernel spent 144 us of a
// C2 step on its one-stage K loop here; the store of 4 x 256 x 256 x 256 floats alone is 50 us).  A wave owns 64 channel
// quads (256 output channels) with their 27 float4 of weights in registers and walks the pixels of its part of an 8 x 32
// pixel tile, four at a time (the 3 x 6 window of a channel is read once per four pixels): lanes = consecutive channels, so a
// store is one contiguous 1 KB line, and the input values are wave-uniform (LDS broadcast reads).  Packed fp32 FMAs
// (2 x 27 per pixel and quad).  The GroupNorm moments of the output (AzConvArgs.gn_quads) cost 10 VALU per pixel: a lane keeps
// its quad for all its pixels (pivoted sums), the four waves of a tile are folded through LDS in a fixed order.
// Because the source is the latent's own layout, the sampling loop needs no NHWC copy of it: the transition kernel writes
// c_in' x_s planar (its flat form, 16 B per element, no pad channel).
constexpr int ST_TH = 8, ST_TW = 32;  // pixel tile of a workgroup
template <int CI>
__global__ __launch_bounds__(256) void conv_stem_kernel(ConvP p, int tiles_w, int tiles_img) {
  __shared__ float xs[CI][ST_TH + 2][ST_TW + 4];
  __shared__ float red[3][4][64];
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / tiles_img;
  const int t = blockIdx.x - b * tiles_img;
  const int th = t / tiles_w, tw = t - th * tiles_w;
  const int H = a.hin, W = a.win;
  for (int e = tid; e < CI * (ST_TH + 2) * (ST_TW + 2); e += 256) {
    const int c = e / ((ST_TH + 2) * (ST_TW + 2));
    const int r2 = e - c * ((ST_TH + 2) * (ST_TW + 2));
    const int r = r2 / (ST_TW + 2), col = r2 - r * (ST_TW + 2);
    const int ih = wrap_coord(th * ST_TH - 1 + r, H, a.pad_mode);
    const int iw = wrap_coord(tw * ST_TW - 1 + col, W, a.pad_mode);
    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    xs[c][r][col] = ok ? a.src0[(((int64_t)b * CI + c) * H + ih) * W + iw] : 0.f;
  }
  __syncthreads();
  const int Q = a.cout_s / 4;
  for (int q0 = 0; q0 < Q; q0 += 64) {
    const int q = q0 + lane;
    const bool live = q < Q;
    f32x2 w[9 * CI][2];
#pragma unroll
    for (int k = 0; k < 9 * CI; ++k) {
      const float4 v = live ? ld4(a.weight + (int64_t)k * a.cout_s + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      w[k][0] = f32x2{v.x, v.y};
      w[k][1] = f32x2{v.z, v.w};
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && live) bv = ld4(a.bias + q * 4);
    float cnt = 0.f, pivot = 0.f, s1 = 0.f, s2 = 0.f;
    for (int g = 0; g < 16; ++g) {  // the wave's rows 2 wave, 2 wave + 1; 8 groups of 4 pixels per row
      const int r = 2 * wave + (g >> 3), c0 = (g & 7) * 4;
      const int oh = th * ST_TH + r;
      if (oh >= H || tw * ST_TW + c0 >= W) continue;  // (wave-uniform)
      f32x2 acc[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j][0] = f32x2{bv.x, bv.y}, acc[j][1] = f32x2{bv.z, bv.w};
#pragma unroll
      for (int c = 0; c < CI; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          float xv[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) xv[i] = xs[c][r + ky][c0 + i];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int k = (ky * 3 + kx) * CI + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x2 x2 = f32x2{xv[j + kx], xv[j + kx]};
              acc[j][0] = __builtin_elementwise_fma(w[k][0], x2, acc[j][0]);
              acc[j][1] = __builtin_elementwise_fma(w[k][1], x2, acc[j][1]);
            }
          }
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ow = tw * ST_TW + c0 + j;
        if (ow >= W || !live) continue;
        float4 o = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
        if (a.act == 1) o = make_float4(az_silu(o.x), az_silu(o.y), az_silu(o.z), az_silu(o.w));
        *reinterpret_cast<float4*>(a.dst + (((int64_t)b * H + oh) * W + ow) * a.cout_s + q * 4) = o;
        if (a.gn_quads != nullptr) {
          if (cnt == 0.f) pivot = o.x;
          const float d0 = o.x - pivot, d1 = o.y - pivot, d2 = o.z - pivot, d3 = o.w - pivot;
          s1 += (d0 + d1) + (d2 + d3);
          s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          cnt += 4.f;
        }
      }
    }
    if (a.gn_quads != nullptr) {  // (n, mean, M2) per lane, then waves 1..3 into wave 0 (Chan, fixed order)
      float mean = 0.f, m2 = 0.f;
      if (cnt > 0.f) {
        mean = pivot + s1 / cnt;
        m2 = s2 - s1 * s1 / cnt;
      }
      __syncthreads();
      red[0][wave][lane] = cnt;
      red[1][wave][lane] = mean;
      red[2][wave][lane] = m2;
      __syncthreads();
      if (wave == 0 && live) {
        for (int k = 1; k < 4; ++k) {
          const float nb = red[0][k][lane], mb = red[1][k][lane], vb = red[2][k][lane];
          if (nb > 0.f) {
            if (cnt > 0.f) {
              const float nn = cnt + nb, d = mb - mean;
              mean = mean + d * (nb / nn);
              m2 = (m2 + vb) + d * d * (cnt * nb / nn);
              cnt = nn;
            } else {
              cnt = nb, mean = mb, m2 = vb;
            }
          }
        }
        float* out = a.gn_quads + (((int64_t)b * tiles_img + t) * Q + q) * 4;
        out[0] = cnt;
        out[1] = mean;
        out[2] = m2;
        out[3] = 0.f;
      }
    }
  }
}

// Split-K combine + epilogue: one thread per (pixel, 4 channels).  IO: element type of dst / res (AzConvArgs.dst_dtype x the entry's type).
template <int IO = 0>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvP p) {
  const AzConvArgs& a = p.a;
  const int q = a.cout_s / 4;
  const int64_t total = (int64_t)p.npix * q;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(e / q);
    const int co = (int)(e - (int64_t)n * q) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < a.splitk; ++z) {
      const float4 w = ld4(a.workspace + ((int64_t)z * p.npix + n) * a.cout_s + co);
      v.x += w.x;
      v.y += w.y;
      v.z += w.z;
      v.w += w.w;
    }
    epilogue_store<IO>(a, n, co, v);
  }
}

// The same combine + epilogue with the GroupNorm moments of the OUTPUT (AzConvArgs.gn_quads; split-K layers are the small
// maps, whose separate statistics launches cost 8 us each -- launch latency, not bandwidth).  One workgroup = one chunk of
// `cpix` pixels of ONE image; thread -> (channel quad q = tid % Q, pixel slot tid / Q): it owns its quad for every pixel it
// visits, so its pivoted sums need no cross-thread traffic until the end, where the (at most 256 / Q) slots of a quad are
// folded with Chan's combination in a fixed order through LDS (deterministic).  Layout of the partials: az_conv2d_winograd_f32's
// ([image][chunk][quad] x (n, mean, M2, 0)), n = the number of values behind a partial (az_groupnorm_finalize_f32 takes any n).
template <int IO = 0>
__global__ __launch_bounds__(256) void conv_splitk_reduce_stats_kernel(ConvP p, int cpix) {
  __shared__ float sh[3 * 256];
  const AzConvArgs& a = p.a;
  const int Q = a.cout_s / 4;
  const int hw = a.hout * a.wout;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int slots = Q >= 256 ? 1 : 256 / Q;
  const int p0 = chunk * cpix, p1 = min(hw, p0 + cpix);
  for (int q0 = 0; q0 < Q; q0 += 256) {  // (Q > 256: every thread owns several quads in turn)
    const int q = q0 + (Q >= 256 ? tid : tid % Q);
    const int slot = Q >= 256 ? 0 : tid / Q;
    float cnt = 0.f, pivot = 0.f, s1 = 0.f, s2 = 0.f;
    if (q < Q && slot < slots) {
      for (int px = p0 + slot; px < p1; px += slots) {
        const int n = b * hw + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int z = 0; z < a.splitk; ++z) {
          const float4 w = ld4(a.workspace + ((int64_t)z * p.npix + n) * a.cout_s + q * 4);
          v.x += w.x;
          v.y += w.y;
          v.z += w.z;
          v.w += w.w;
        }
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f), r = g, bv = g;
        if (a.bias) bv = ld4(a.bias + q * 4);
        epilogue_fetch<IO>(a, n, b, q * 4, g, r);
        const float4 f = epilogue_apply_store<IO>(a, n, b, q * 4, v, bv, g, r);
        if (cnt == 0.f) pivot = f.x;
        const float d0 = f.x - pivot, d1 = f.y - pivot, d2 = f.z - pivot, d3 = f.w - pivot;
        s1 += (d0 + d1) + (d2 + d3);
        s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        cnt += 4.f;
      }
    }
    // (n, mean, M2) of this thread's values; then the slots of a quad, in slot order
    float mean = 0.f, m2 = 0.f;
    if (cnt > 0.f) {
      mean = pivot + s1 / cnt;
      m2 = s2 - s1 * s1 / cnt;
    }
    if (slots > 1) {
      __syncthreads();
      sh[tid] = cnt;
      sh[256 + tid] = mean;
      sh[512 + tid] = m2;
      __syncthreads();
      if (slot == 0 && q < Q) {
        for (int k = 1; k < slots; ++k) {
          const float nb = sh[k * Q + q], mb = sh[256 + k * Q + q], vb = sh[512 + k * Q + q];
          if (nb > 0.f) {
            const float nn = cnt + nb, d = mb - mean;
            mean = mean + d * (nb / nn);
            m2 = (m2 + vb) + d * d * (cnt * nb / nn);
            cnt = nn;
          }
        }
      }
    }
    if (slot == 0 && q < Q) {
      float* out = a.gn_quads + (((int64_t)b * a.gn_chunks + chunk) * Q + q) * 4;
      out[0] = cnt;
      out[1] = mean;
      out[2] = m2;
      out[3] = 0.f;
    }
  }
}

// =================================================================================================
// Winograd F(2x2, 3x3) form of the stride-1 3x3 convolution, fully fused:
//   out tile (2x2) = A^T [ sum_ci U[xi,nu][co][ci] * V[xi,nu][ci] ] A ,  U = G g G^T (offline),  V = B^T d B
// 16 "frequency" GEMMs of depth Cin replace the 9-tap GEMM: 4 multiplies per output instead of 9
// (2.25x fewer MFMA passes) in exact fp32 arithmetic (the transforms only add / subtract; the 1/2
// factors live in the offline filter transform).  One kernel does everything:
//   * the input transform B^T d B runs in the loader threads on the raw 4x4 patches (gathered
//     with the same bounds-checked buffer loads as the direct kernel: padding, two sources and
//     nearest-x2 upsampling are address arithmetic);
//   * 8 waves (2 per SIMD, so one wave's MFMAs cover its SIMD partner's loads / transform / barrier
//     waits): each owns 8 of the 16 frequencies of a (32 couts x 32 tiles) sub-block as 8
//     accumulators of v_mfma_f32_32x32x2_f32; the output transform A^T M A is linear, so every wave
//     reduces its 8 frequencies to a partial 2x2 output in registers; both halves park it in an LDS
//     exchange buffer laid out [tile][pixel][cout] and all 8 waves read it back row-wise, so that the
//     fused epilogue (bias, SiLU, gate, residual, store) writes whole 128-byte lines.
// Block = 64 couts x 64 tiles (= 256 output pixels), K stage = 8 input channels, XOR-swizzled
// 8-float LDS rows (conflict-free ds_read_b128 fragments), 2 stages = 128 KB (131 KB with the padded
// exchange buffer that reuses them) -> one workgroup per CU.
constexpr int WT = 64;                 // tiles per workgroup (= 256 output pixels)
constexpr int WC = 64;                 // output channels per workgroup
constexpr int WK = 8;                  // input channels per stage
constexpr int WU_STAGE = 16 * WC * WK; // floats
constexpr int WV_STAGE = 16 * WT * WK;
constexpr int W_STAGE = WU_STAGE + WV_STAGE;  // 16384 floats = 64 KB; two stages = 128 KB
constexpr int W_OT = 4 * WC + 4;              // epilogue: floats per tile row of the [tile][pixel][cout] exchange buffer
constexpr int W_LDS_BYTES = (2 * WT * W_OT + 3 * WT + 384) * 4;  // 135,424 B >= the two K-loop stages (+384: GroupNorm partials)



// LDS rows hold 8 floats (one 8-channel chunk) = two 16-byte halves; the half index is XORed with
// bit 3 of the row so that the ds_read_b128 fragment reads (32 rows x 1 half per half-wave) and the
// ds_write_b128 of the loaders are bank-conflict-free without padding.
__device__ __forceinline__ int wswz(int row, int half) { return row * WK + 4 * (half ^ ((row >> 3) & 1)); }

#include "wino_kloop.inc"
constexpr int W_VOFF_BYTES = 256 * 64;  // ASM form: the second source's 16 patch offsets of the 256 gather threads

// ASM = true: the K loop is the hand-scheduled instruction stream of wino_kloop.inc (gen_wino_kloop.py), software
// pipelined across the barrier; ASM = false: the C++ loop (one barrier per stage behind the stores), kept for A/B runs
// (AZ_WINOGRAD_ASM=0).  Prologue address arithmetic and the whole epilogue are shared.
template <bool ASM>
__global__ __launch_bounds__(512, 2) void conv_winograd_kernel(WinoP p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // 2 * W_STAGE floats
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;        // 8 waves = 2 per SIMD: wave w and w+4 share a SIMD
  const int fh = wave >> 2;         // which 8 of the 16 frequencies (xi in {0,1} or {2,3})
  const int wco = (wave >> 1) & 1;  // which 32 of the 64 couts
  const int wti = wave & 1;         // which 32 of the 64 tiles
  const int l31 = lane & 31;
  const int h = lane >> 5;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  // An XCD runs 32 workgroups at a time out of its contiguous range of `wg`: make them a RECTANGLE of gt tile blocks x gc cout
  // blocks (host: 8 x 4 where the grid allows), so that its L2 fetches 8 tile blocks' patches and 4 cout blocks' filter chunks
  // per stage instead of 2 and 16 on the 1024-channel layers (a filter chunk is 2 - 4 x the bytes of a tile block's patches).
  const int rsz = p.gt * p.gc;
  const int rect = wg / rsz, rin = wg - rect * rsz;
  const int rcols = p.cblocks / p.gc;
  const int rrow = rect / rcols;
  const int rin_c = rin / p.gt;
  const int tb = rrow * p.gt + (rin - rin_c * p.gt);
  const int cb = (rect - rrow * rcols) * p.gc + rin_c;
  const int t0 = tb * WT;
  const int tiles_img = p.tiles_h * p.tiles_w;
  const int b_first = t0 / tiles_img;

  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);

  const int64_t s0_elems = (int64_t)a.h0 * a.w0 * a.c0s;
  const int64_t s1_elems = (int64_t)a.h1 * a.w1 * a.c1s;
  // (depth taps: per-lane source planes are taken relative to b_base; lanes whose plane is masked may land anywhere)
  const int b_base = az_depth_base(a, b_first);
  auto clamp_bytes = [](int64_t e) { return (unsigned)(e <= 0 ? 0 : (e * 4 > AZ_RSRC_CLAMP ? AZ_RSRC_CLAMP : e * 4)); };  // (e <= 0: a depth shift past the last plane)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.weight, 0, clamp_bytes((int64_t)p.nk * p.cblocks * WU_STAGE), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src0 + b_base * s0_elems), 0, clamp_bytes((a.batch - b_base) * s0_elems), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 + b_base * s1_elems : a.src0), 0,
      a.src1 ? clamp_bytes((a.batch - b_base) * s1_elems) : 0u, 0x00020000);

  // ---- loader roles: waves 0..3 (threads 0..255) gather + transform one (tile, channel pair) each
  //      (16 x 8-byte loads); waves 4..7 stream the pre-transformed filter chunk (8 x 16 B each).
  //      Wave w (V role) and wave w+4 (U role) share a SIMD, so every SIMD carries the same load.
  const bool vrole = tid < 256;
  const int vj = (tid & 255) >> 2;  // tile within the block
  const int vq = tid & 3;           // which channel pair of the 8
  int v_b = -1, v_bs = 0, v_ih0 = 0, v_iw0 = 0;  // v_bs: the tile's source plane relative to b_base
  bool v_dok = true;  // AzConvArgs.depth: the source plane (image + depth_shift) lies inside the image's volume
  if (vrole) {
    const int t = t0 + vj;
    if (t < p.ntiles) {
      const int b = t / tiles_img;
      const int r = t - b * tiles_img;
      const int th = r / p.tiles_w;
      v_b = b - b_first;
      v_ih0 = 2 * th - 1;
      v_iw0 = 2 * (r - th * p.tiles_w) - 1;
      v_bs = az_depth_plane(a, b, v_dok) - b_base;
    }
  }
  unsigned voffV[16];
  int cur_src = -1;
  auto set_src = [&](int src) {
    cur_src = src;
    const int cs = src ? a.c1s : a.c0s;
    const int up = src ? a.up1 : a.up0;
    const int hs = src ? a.h1 : a.h0;
    const int ws = src ? a.w1 : a.w0;
    // the patch address is separable: 4 row parts + 4 column parts (8 wraps / bounds tests, not 32 -- the prologue of a
    // workgroup was ~1400 instructions of this, 3 % of a Cin = 256 layer)
    int rpart[4], cpart[4];
    bool rok[4], cok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ih = wrap_coord(v_ih0 + r, a.hin, a.pad_mode);
      rok[r] = v_b >= 0 && v_dok && (unsigned)ih < (unsigned)a.hin;
      rpart[r] = (v_bs * hs + (ih >> up)) * ws * cs * 4;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int iw = wrap_coord(v_iw0 + c, a.win, a.pad_mode);
      cok[c] = (unsigned)iw < (unsigned)a.win;
      cpart[c] = ((iw >> up) * cs + vq * 2) * 4;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) voffV[r * 4 + c] = rok[r] && cok[c] ? (unsigned)(rpart[r] + cpart[c]) : OOB;
  };

  f32x2 rv[16];  // V role: raw 4x4 patch of a channel pair; U role: 8 filter float4 (rv[2i], rv[2i+1])

  int staged_kt = 0;  // (C++ loop) the stage whose raw patch sits in rv
  auto load_stage = [&](int kt) {  // kt is wave-uniform: descriptor choice stays provably uniform
    const bool src1 = kt >= p.nkc0;
    staged_kt = kt;
    if (vrole) {
      if ((src1 ? 1 : 0) != cur_src) set_src(src1 ? 1 : 0);
      const int kc = src1 ? kt - p.nkc0 : kt;
      const unsigned soff = (unsigned)(kc * WK * 4);
      const int cs = src1 ? a.c1s : a.c0s;
      if (kc * WK + WK <= cs) {  // whole chunk inside the source (wave-uniform): no per-lane channel-tail select
        if (src1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) rv[i] = buf_ld2(rs1, voffV[i], soff);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) rv[i] = buf_ld2(rs0, voffV[i], soff);
        }
      } else {
        const bool kv = kc * WK + vq * 2 < cs;
        if (src1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) rv[i] = buf_ld2(rs1, kv ? voffV[i] : OOB, soff);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) rv[i] = buf_ld2(rs0, kv ? voffV[i] : OOB, soff);
        }
      }
    } else {
      const unsigned soff = (unsigned)(((int64_t)kt * p.cblocks + cb) * (WU_STAGE * 4));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = buf_ld4(rw, (unsigned)((tid - 256 + 256 * i) * 16), soff);
        rv[2 * i] = f32x2{v.x, v.y};
        rv[2 * i + 1] = f32x2{v.z, v.w};
      }
    }
  };

  const int voffL = wswz(vj, vq >> 1) + 2 * (vq & 1);  // + f * WT * WK   (V, frequency f)
  auto store_stage = [&](int buf) {
    float* Us = wsm + buf * W_STAGE;
    float* Vs = Us + WU_STAGE;
    if (vrole) {
      if (a.in_affine != nullptr) {
        // the input is act(x * scale + shift) (the GroupNorm apply pass, fused); padding positions stay zero
        const float* sp = a.in_affine + ((int64_t)(b_first + (v_b < 0 ? 0 : v_b)) * a.c0s + staged_kt * WK + vq * 2);
        const f32x2 sc = *reinterpret_cast<const f32x2*>(sp);
        const f32x2 sh = *reinterpret_cast<const f32x2*>(sp + (int64_t)a.batch * a.c0s);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          f32x2 v = rv[i] * sc + sh;
          if (a.in_act == 1) v = f32x2{az_silu(v.x), az_silu(v.y)};
          rv[i] = voffV[i] == OOB ? f32x2{0.f, 0.f} : v;
        }
      }
      // in-place V = B^T d B (packed fp32 adds); B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x2 d0 = rv[c], d1 = rv[4 + c], d2 = rv[8 + c], d3 = rv[12 + c];
        rv[c] = d0 - d2;
        rv[4 + c] = d1 + d2;
        rv[8 + c] = d2 - d1;
        rv[12 + c] = d1 - d3;
      }
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const f32x2 u0 = rv[4 * xi], u1 = rv[4 * xi + 1], u2 = rv[4 * xi + 2], u3 = rv[4 * xi + 3];
        *reinterpret_cast<f32x2*>(Vs + (4 * xi + 0) * WT * WK + voffL) = u0 - u2;
        *reinterpret_cast<f32x2*>(Vs + (4 * xi + 1) * WT * WK + voffL) = u1 + u2;
        *reinterpret_cast<f32x2*>(Vs + (4 * xi + 2) * WT * WK + voffL) = u2 - u1;
        *reinterpret_cast<f32x2*>(Vs + (4 * xi + 3) * WT * WK + voffL) = u1 - u3;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid - 256 + 256 * i;  // float4 index in the [16][64][8] chunk: row = e >> 1, half = e & 1
        *reinterpret_cast<float4*>(Us + wswz(e >> 1, e & 1)) = make_float4(rv[2 * i].x, rv[2 * i].y, rv[2 * i + 1].x, rv[2 * i + 1].y);
      }
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  // fragment addresses: row (within a frequency) = wave's 32-row block + l31; half = h (swizzled)
  const int fragA = (fh * 8) * WC * WK + wswz(wco * 32 + l31, h);  // + f * WC * WK
  const int fragB = (fh * 8) * WT * WK + wswz(wti * 32 + l31, h);  // + f * WT * WK
  if constexpr (ASM) {
    if (kt_begin < kt_end) {
      typedef __attribute__((address_space(3))) float lds_float;
      const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)wsm;
      const unsigned fragA_b = lds0 + (unsigned)fragA * 4u;
      const unsigned fragB_b = lds0 + (unsigned)(WU_STAGE + fragB) * 4u;
      if (vrole) {
        // patch offsets of the slice's first source -> LDS (parked in stage buffer 1, which nobody writes before the
        // first barrier), of the second source (if the slice crosses into it) -> the area behind the stages
        const bool start1 = kt_begin >= p.nkc0;
        const int kt_switch = (!start1 && kt_end > p.nkc0) ? p.nkc0 : 0x7fffffff;
        set_src(start1 ? 1 : 0);
        uint4* pa = reinterpret_cast<uint4*>(reinterpret_cast<char*>(wsm) + W_STAGE * 4 + tid * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) pa[q] = make_uint4(voffV[4 * q], voffV[4 * q + 1], voffV[4 * q + 2], voffV[4 * q + 3]);
        if (kt_switch != 0x7fffffff) {
          set_src(1);
          uint4* pb = reinterpret_cast<uint4*>(reinterpret_cast<char*>(wsm) + W_LDS_BYTES + tid * 64);
#pragma unroll
          for (int q = 0; q < 4; ++q) pb[q] = make_uint4(voffV[4 * q], voffV[4 * q + 1], voffV[4 * q + 2], voffV[4 * q + 3]);
        }
        const unsigned ldsA = lds0 + W_STAGE * 4 + tid * 64, ldsB = lds0 + W_LDS_BYTES + tid * 64;
        const unsigned vst = lds0 + (unsigned)(WU_STAGE + voffL) * 4u;
        // buffer descriptors as words (the asm cannot address the halves of a 128-bit operand)
        const uint64_t b0 = (uint64_t)(uintptr_t)(a.src0 + b_base * s0_elems);
        const uint64_t b1 = (uint64_t)(uintptr_t)(a.src1 ? a.src1 + b_base * s1_elems : a.src0);
        const unsigned n0 = clamp_bytes((a.batch - b_base) * s0_elems), n1 = a.src1 ? clamp_bytes((a.batch - b_base) * s1_elems) : 0u;
        unsigned d0w0 = (unsigned)b0, d0w1 = (unsigned)(b0 >> 32) & 0xffffu, d0w2 = n0;
        unsigned d1w0 = (unsigned)b1, d1w1 = (unsigned)(b1 >> 32) & 0xffffu, d1w2 = n1;
        if (start1) d0w0 = d1w0, d0w1 = d1w1, d0w2 = d1w2;
        const unsigned dflags = 0x00020000u;
        const int soff0 = (start1 ? kt_begin - p.nkc0 : kt_begin) * (WK * 4);
        const int tail0 = (a.c0s & (WK - 1)) ? p.nkc0 - 1 : -1;
        const int tail1 = (a.c1s & (WK - 1)) ? p.nk - 1 : -1;
        // in_affine: [scale | shift], (batch, c0s) each; per-lane offset of the thread's image and channel pair
        const uint64_t baf = (uint64_t)(uintptr_t)a.in_affine;
        const unsigned af_lo = (unsigned)baf, af_hi = (unsigned)(baf >> 32);
        const unsigned af_delta = a.in_affine ? (unsigned)(a.batch * a.c0s * 4) : 0u;
        const unsigned af_voff = (unsigned)(((b_first + (v_b < 0 ? 0 : v_b)) * a.c0s + vq * 2) * 4);
        if (a.in_affine != nullptr) {  // (a second stream: the plain one carries no branch for this)
          asm volatile(WINO_KLOOP_VA_ASM
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                       : "v"(fragA_b), "v"(fragB_b), "v"(vst), "v"(ldsA), "v"(ldsB), "s"(d0w0), "s"(d0w1), "s"(d0w2), "s"(dflags),
                         "s"(d1w0), "s"(d1w1), "s"(d1w2), "s"(dflags), "s"(kt_begin), "s"(kt_end), "s"(kt_switch), "s"(soff0),
                         "s"(tail0), "s"(tail1), "v"(af_voff), "s"(af_lo), "s"(af_hi), "s"(af_delta)
                       : WINO_KLOOP_VA_CLOBBERS);
        } else {
          asm volatile(WINO_KLOOP_V_ASM
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                       : "v"(fragA_b), "v"(fragB_b), "v"(vst), "v"(ldsA), "v"(ldsB), "s"(d0w0), "s"(d0w1), "s"(d0w2), "s"(dflags),
                         "s"(d1w0), "s"(d1w1), "s"(d1w2), "s"(dflags), "s"(kt_begin), "s"(kt_end), "s"(kt_switch), "s"(soff0),
                         "s"(tail0), "s"(tail1), "v"(af_voff), "s"(af_lo), "s"(af_hi), "s"(af_delta)
                       : WINO_KLOOP_CLOBBERS);
        }
      } else {
        const uint64_t bw = (uint64_t)(uintptr_t)a.weight;
        const unsigned ww0 = (unsigned)bw, ww1 = (unsigned)(bw >> 32) & 0xffffu;
        const unsigned ww2 = clamp_bytes((int64_t)p.nk * p.cblocks * WU_STAGE), ww3 = 0x00020000u;
        const int e0 = tid - 256;
        const unsigned uvoff = (unsigned)(tid - 256) * 16u;
        const unsigned ust = lds0 + (unsigned)wswz(e0 >> 1, e0 & 1) * 4u;
        const unsigned usoff0 = (unsigned)(((int64_t)kt_begin * p.cblocks + cb) * (WU_STAGE * 4));
        const unsigned ustep = (unsigned)(p.cblocks * (WU_STAGE * 4));
        asm volatile(WINO_KLOOP_U_ASM
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(fragA_b), "v"(fragB_b), "v"(ust), "v"(uvoff), "s"(ww0), "s"(ww1), "s"(ww2), "s"(ww3), "s"(kt_begin), "s"(kt_end),
                       "s"(usoff0), "s"(ustep)
                     : WINO_KLOOP_CLOBBERS);
      }
    } else {
      __syncthreads();
    }
  } else {
  if (kt_begin < kt_end) {
    load_stage(kt_begin);
    store_stage(0);
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    // Code placement: this loop loses 4.5 - 7 % when its body starts at byte phases 16..23 of a 32-byte window (r02:
    // swept with s_nop padding -- every other phase is equally fast; an innocuous edit of the PROLOGUE, even code that
    // never executes, moved it there twice).  The anchor sits INSIDE the loop, so the phase of the body no longer depends
    // on what precedes the loop; its padding (<= 15 s_nop per stage) is noise.  2 s_nop = the fastest phase of the sweep.
    asm volatile(".p2align 6\n s_nop 0\n s_nop 0" ::: "memory");
    const int buf = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) load_stage(kt + 1);  // in flight under this stage's MFMAs (and the partner wave's)
    const float* Us = wsm + buf * W_STAGE;
    const float* Vs = Us + WU_STAGE;
    float4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const float4*>(Us + fragA);
    fb[0] = *reinterpret_cast<const float4*>(Vs + fragB);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int cur = f & 1, nxt = cur ^ 1;
      if (f < 7) {
        fa[nxt] = *reinterpret_cast<const float4*>(Us + (f + 1) * WC * WK + fragA);
        fb[nxt] = *reinterpret_cast<const float4*>(Vs + (f + 1) * WT * WK + fragB);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur].z, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur].w, acc[f], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
  }
  // ---- output transform.  Y = A^T M A is linear in M, so each wave transforms the 8 frequencies
  // (two xi rows) it owns into a partial 2x2 output; the xi in {2,3} waves hand theirs to their
  // xi in {0,1} partners through LDS (once per workgroup), which add and run the fused epilogue.
  // Lane: tile = wti*32 + l31; couts wco*32 + 8*g + 4*h + (0..3) in registers 4g .. 4g+3.
  // (Two code versions behind a scalar branch on the wave-uniform frequency half, and explicit register PAIRS: the
  // compiler had if-converted the two cases into 128 v_cndmask around both results and kept every add scalar -- ~450 vector
  // instructions per thread where 96 packed ones do; vector instructions are not hidden on this chip, section 7c.)
  float y[4][4][4];  // [g][pixel py*2+px][r]
  auto out_transform = [&](auto FH) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        f32x2 s[2][4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          const f32x2 ma = {acc[nu][4 * g + 2 * k], acc[nu][4 * g + 2 * k + 1]};          // xi = 2*fh
          const f32x2 mb = {acc[4 + nu][4 * g + 2 * k], acc[4 + nu][4 * g + 2 * k + 1]};  // xi = 2*fh + 1
          if constexpr (decltype(FH)::value == 0) {
            s[0][nu] = ma + mb;  // A^T rows: [1 1 1 0], [0 1 -1 -1]
            s[1][nu] = mb;
          } else {
            s[0][nu] = ma;
            s[1][nu] = -ma - mb;
          }
        }
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const f32x2 y0 = (s[py][0] + s[py][1]) + s[py][2];
          const f32x2 y1 = (s[py][1] - s[py][2]) - s[py][3];
          y[g][py * 2 + 0][2 * k] = y0.x;
          y[g][py * 2 + 0][2 * k + 1] = y0.y;
          y[g][py * 2 + 1][2 * k] = y1.x;
          y[g][py * 2 + 1][2 * k + 1] = y1.y;
        }
      }
  };
  if (fh == 0) {
    asm volatile("" ::: "memory");  // (keeps the two versions apart: no if-conversion)
    out_transform(std::integral_constant<int, 0>{});
  } else {
    asm volatile("" ::: "memory");
    out_transform(std::integral_constant<int, 1>{});
  }
  // Both frequency halves park their partial outputs in LDS as [tile][pixel][cout] (tile stride padded by 4 floats:
  // conflict-free ds_write_b128), then ALL 8 waves read rows back so that 16 consecutive lanes store the 256 contiguous
  // bytes of one output pixel (two full 128-byte lines) instead of 32 bytes of 32 different pixels per instruction.
  float* obuf = wsm + fh * (WT * W_OT);
  {
    float* orow = obuf + (wti * 32 + l31) * W_OT + wco * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int px = 0; px < 4; ++px)
        *reinterpret_cast<float4*>(orow + px * WC + 8 * g) = make_float4(y[g][px][0], y[g][px][1], y[g][px][2], y[g][px][3]);
  }
  int* tinfo = reinterpret_cast<int*>(wsm + 2 * WT * W_OT);  // [64] first output pixel of the tile (-1: none), [64] flags, [64] image
  if (wave < 2 && h == 0) {
    const int t = t0 + wti * 32 + l31;
    int n00 = -1, fl = 0, b = 0;
    if (t < p.ntiles) {
      b = t / tiles_img;
      const int rr = t - b * tiles_img;
      const int th = rr / p.tiles_w;
      const int tw = rr - th * p.tiles_w;
      n00 = (b * a.hout + 2 * th) * a.wout + 2 * tw;
      fl = (2 * th + 1 < a.hout ? 1 : 0) | (2 * tw + 1 < a.wout ? 2 : 0);
    }
    tinfo[wti * 32 + l31] = n00;
    tinfo[WT + wti * 32 + l31] = fl;
    tinfo[2 * WT + wti * 32 + l31] = b;
  }
  __syncthreads();
  const int cq = tid & 15;  // the same channel quad in every iteration
  const int co = cb * WC + cq * 4;
  if (co >= a.cout_s) return;
  int on[8], ob[8];
  float4 ov[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = (it * 512 + tid) >> 4;  // tile * 4 + pixel
    const int tile = row >> 2, px = row & 3;
    const int n00 = tinfo[tile], fl = tinfo[WT + tile];
    ob[it] = tinfo[2 * WT + tile];
    const bool skip = n00 < 0 || (((px >> 1) & ~fl) | ((px & 1) & ~(fl >> 1)));
    on[it] = skip ? -1 : n00 + (px >> 1) * a.wout + (px & 1);
    const float4 v0 = *reinterpret_cast<const float4*>(wsm + tile * W_OT + px * WC + cq * 4);
    const float4 v1 = *reinterpret_cast<const float4*>(wsm + WT * W_OT + tile * W_OT + px * WC + cq * 4);
    ov[it] = make_float4(v0.x + v1.x, v0.y + v1.y, v0.z + v1.z, v0.w + v1.w);
  }
  if (a.gn_quads == nullptr) {
    epilogue_store_batch<8>(a, on, ob, co, ov, (int64_t)blockIdx.y * p.npix);
    return;
  }
  // ---- GroupNorm statistics of the OUTPUT, for the normalisation that consumes it (the separate statistics pass read
  // the whole tensor again: 0.38 ms of a C2 step, 10 ms of a C4 step).  The host enables this only when the 64 tiles of
  // a workgroup lie in one image and cout_s % 64 == 0: every thread owns one channel quad over 8 of the block's 256
  // pixels; (n, mean, M2) of those 32 values, Chan-combined over the 32 threads of the quad in a fixed order
  // (deterministic), give one partial per (image, tile block, channel quad) that az_groupnorm_finalize_f32 folds.
  float mom[3] = {0.f, 0.f, 0.f};
  epilogue_store_batch<8, true>(a, on, ob, co, ov, (int64_t)blockIdx.y * p.npix, mom);
  // 32 values per thread (the host admits no ragged tiles here): mean and M2 from the pivoted sums, then Chan's pairwise
  // combination with EQUAL counts as a fixed tree (deterministic) -- lanes cq + 16 j of a wave by two cross-lane steps,
  // the 8 waves through LDS past the tile table (no barrier before the writes).
  float am = mom[0] + mom[1] * (1.f / 32.f);
  float a2 = mom[2] - mom[1] * mom[1] * (1.f / 32.f);
  auto chan = [](float& am, float& a2, float bm, float b2, float half_n) {  // both sides hold 2 * half_n values
    const float d = bm - am;
    am = am + 0.5f * d;
    a2 = (a2 + b2) + d * d * half_n;
  };
  chan(am, a2, __shfl_xor(am, 16), __shfl_xor(a2, 16), 16.f);
  chan(am, a2, __shfl_xor(am, 32), __shfl_xor(a2, 32), 32.f);
  float* sh = reinterpret_cast<float*>(tinfo + 3 * WT);
  if (lane < 16) {
    sh[wave * 16 + lane] = am;
    sh[128 + wave * 16 + lane] = a2;
  }
  __syncthreads();
  if (tid < 16) {
    float wm[8], w2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      wm[k] = sh[k * 16 + tid];
      w2[k] = sh[128 + k * 16 + tid];
    }
    float hn = 64.f;  // each wave's partial: 128 values
#pragma unroll
    for (int o = 1; o < 8; o <<= 1, hn *= 2.f)
#pragma unroll
      for (int k = 0; k < 8; k += 2 * o) chan(wm[k], w2[k], wm[k + o], w2[k + o], hn);
    const float an = 1024.f;
    am = wm[0], a2 = w2[0];
    const int chunk = (t0 - b_first * tiles_img) / WT;
    float* out = a.gn_quads + ((((int64_t)b_first * a.gn_chunks + chunk) * (a.cout_s / 4)) + (cb * (WC / 4) + tid)) * 4;
    out[0] = an;
    out[1] = am;
    out[2] = a2;
    out[3] = 0.f;
  }
}

// =================================================================================================
// Winograd F(4x4, 3x3): 36 frequency GEMMs per 6x6 input patch / 4x4 output tile, i.e. 2.25 multiplies per
// output instead of 9 (direct) or 4 (F(2x2,3x3)).  The transforms now contain the constants 2, 4, 5, 8 and
// the filter transform 1/4 .. 1/24, so this form is NOT exact: its fp32 rounding error is ~20x that of the
// F(2x2) kernel (still ~1e-5 of the output scale per layer).  It is an opt-in
// (AZ_WINOGRAD=4) for that reason -- and because, as measured in round 1, it is not yet faster than the
// F(2x2) kernel at the 256^2 level (both 203 TF/s algorithmic; +5 % at 128^2, +12 % at 64^2): with 18 MFMAs per
// wave per stage the U/V staging (54 KB of loads, 72 KB of LDS writes, one barrier) is exposed.  Ablations
// on 4x256x256x256->256: 1536 us; without the gather 1209, without the filter loads 1225, without either
// 1076, without the two transform passes 1047, without the MFMAs 1192 (MFMA-bound time would be ~570).
//
// Block = 64 couts x 32 tiles (= 512 output pixels), K stage = 4 input channels, 8 waves.  Wave (c2, I, J)
// owns the 3 x 3 frequency block xi in 3I..3I+2, nu in 3J..3J+2 of its 32-cout half as 9 accumulators
// (144 VGPRs).  Per stage: U chunk 36 KB (plain copy of the pre-transformed filter) + V 18 KB.
// The input transform is done by all threads in two 1-D passes through an LDS scratch so that no thread ever
// holds a whole 6x6 patch:
//   pass A (after the MFMAs of stage s):   thread (tile, patch row r, channel pair) has loaded its 6 pixels
//       (issued one iteration earlier), applies B^T along the row and writes 6 values to the scratch;
//   pass B (before the MFMAs of stage s+1): thread (tile, nu, channel pair) reads the 6 rows of column nu,
//       applies B^T down the column and writes the 6 frequencies (xi, nu) into the V stage buffer.
// LDS: 2 x (36 + 18) KB stages + 2 x 18 KB scratch = 144 KB -> one workgroup per CU.
constexpr int W4T = 32;                  // tiles per workgroup
constexpr int W4C = 64;                  // couts per workgroup
constexpr int W4K = 4;                   // input channels per stage
constexpr int W4U_F = W4C * W4K;         // floats per frequency, U
constexpr int W4V_F = W4T * W4K + 8;     // floats per frequency, V (+8: spreads pass-B writes over the banks)
constexpr int W4U_STAGE = 36 * W4U_F;    // 9216 floats
constexpr int W4V_STAGE = 36 * W4V_F;    // 4896 floats
constexpr int W4_STAGE = W4U_STAGE + W4V_STAGE;
constexpr int W4_SCRATCH = W4T * 6 * 2 * 6 * 2;  // 4608 floats: [tile][nu][pair][row] f32x2
constexpr int W4_LDS = 2 * W4_STAGE + 2 * W4_SCRATCH;  // 37440 floats = 146.25 KB

struct Wino4P {
  AzConvArgs a;
  int npix;
  int tiles_h, tiles_w, ntiles;
  int nkc0, nk;   // 4-channel chunks of source 0, total
  int kps;        // chunks per split
  int cblocks;
  int tblocks;
};

// 1-D input transform B^T d, points (0, 1, -1, 2, -2, inf).
template <class T>
__device__ __forceinline__ void wino4_bt(const T* d, T* o) {
  const T p = d[4] - 4.f * d[2];
  const T q = d[3] - 4.f * d[1];
  const T r = d[4] - d[2];
  const T s = d[3] - d[1];
  o[0] = 4.f * (d[0] - d[2]) + r;
  o[1] = p + q;
  o[2] = p - q;
  o[3] = r + 2.f * s;
  o[4] = r - 2.f * s;
  o[5] = (d[5] - d[3]) - 4.f * s;
}

// Partial 1-D output transform: contribution of m[3J .. 3J+2] to A^T m,
// A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].
__device__ __forceinline__ void wino4_at_part(int J, float m0, float m1, float m2, float* y) {
  if (J == 0) {  // columns 0, 1, 2
    const float s = m1 + m2, d = m1 - m2;
    y[0] = m0 + s;
    y[1] = d;
    y[2] = s;
    y[3] = d;
  } else {  // columns 3, 4, 5
    const float s = m0 + m1, d = m0 - m1;
    y[0] = s;
    y[1] = 2.f * d;
    y[2] = 4.f * s;
    y[3] = 8.f * d + m2;
  }
}

__global__ __launch_bounds__(512, 2) void conv_winograd4_kernel(Wino4P p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c2 = wave & 1;          // 32-cout half
  const int gI = (wave >> 1) & 1;   // xi block
  const int gJ = wave >> 2;         // nu block
  const int l31 = lane & 31;
  const int h = lane >> 5;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tb = wg / p.cblocks;
  const int cb = wg - tb * p.cblocks;
  const int t0 = tb * W4T;
  const int tiles_img = p.tiles_h * p.tiles_w;
  const int b_first = t0 / tiles_img;

  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);
  const int nst = kt_end - kt_begin;

  const int64_t s0_elems = (int64_t)a.h0 * a.w0 * a.c0s;
  const int64_t s1_elems = (int64_t)a.h1 * a.w1 * a.c1s;
  auto clamp_bytes = [](int64_t e) { return (unsigned)(e <= 0 ? 0 : (e * 4 > AZ_RSRC_CLAMP ? AZ_RSRC_CLAMP : e * 4)); };  // (e <= 0: a depth shift past the last plane)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.weight, 0, clamp_bytes((int64_t)p.nk * p.cblocks * W4U_STAGE), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src0 + b_first * s0_elems), 0, clamp_bytes((a.batch - b_first) * s0_elems), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 + b_first * s1_elems : a.src0), 0,
      a.src1 ? clamp_bytes((a.batch - b_first) * s1_elems) : 0u, 0x00020000);

  // ---- pass A role: thread -> (tile, patch row, channel pair); 384 of the 512 threads
  const bool arole = tid < 384;
  const int a_tile = tid / 12;
  const int a_row = (tid - a_tile * 12) >> 1;
  const int a_q = tid & 1;
  int v_b = -1, v_ih = 0, v_iw0 = 0;
  if (arole) {
    const int t = t0 + a_tile;
    if (t < p.ntiles) {
      const int b = t / tiles_img;
      const int r = t - b * tiles_img;
      const int th = r / p.tiles_w;
      v_b = b - b_first;
      v_ih = 4 * th - 1 + a_row;
      v_iw0 = 4 * (r - th * p.tiles_w) - 1;
    }
  }
  unsigned voffV[6];
  int cur_src = -1;
  auto set_src = [&](int src) {
    cur_src = src;
    const int cs = src ? a.c1s : a.c0s;
    const int up = src ? a.up1 : a.up0;
    const int hs = src ? a.h1 : a.h0;
    const int ws = src ? a.w1 : a.w0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int iw = wrap_coord(v_iw0 + c, a.win, a.pad_mode);
      const int ihw = wrap_coord(v_ih, a.hin, a.pad_mode);
      const bool ok = v_b >= 0 && (unsigned)ihw < (unsigned)a.hin && (unsigned)iw < (unsigned)a.win;
      const int pix = (v_b * hs + (ihw >> up)) * ws + (iw >> up);
      voffV[c] = ok ? (unsigned)((pix * cs + a_q * 2) * 4) : OOB;
    }
  };

  f32x2 rawv[6];   // pass A: this thread's 6 pixels of one patch row (channel pair)
  float4 rawu[5];  // filter chunk: float4 #(tid + 512 j)

  auto load_v = [&](int kt) {  // kt wave-uniform
    if (!arole) return;
    const bool src1 = kt >= p.nkc0;
    if ((src1 ? 1 : 0) != cur_src) set_src(src1 ? 1 : 0);
    const unsigned soff = (unsigned)((src1 ? kt - p.nkc0 : kt) * W4K * 4);
    if (src1) {
#pragma unroll
      for (int c = 0; c < 6; ++c) rawv[c] = buf_ld2(rs1, voffV[c], soff);
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) rawv[c] = buf_ld2(rs0, voffV[c], soff);
    }
  };
  auto load_u = [&](int kt) {
    const unsigned soff = (unsigned)(((int64_t)kt * p.cblocks + cb) * (W4U_STAGE * 4));
#pragma unroll
    for (int j = 0; j < 4; ++j) rawu[j] = buf_ld4(rw, (unsigned)((tid + 512 * j) * 16), soff);
    rawu[4] = buf_ld4(rw, tid < 256 ? (unsigned)((tid + 2048) * 16) : OOB, soff);
  };
  auto store_u = [&](int buf) {
    float* Us = wsm + buf * W4_STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(Us + (tid + 512 * j) * 4) = rawu[j];
    if (tid < 256) *reinterpret_cast<float4*>(Us + (tid + 2048) * 4) = rawu[4];
  };
  // scratch slot of (tile, nu, pair) = 12 floats at 12 * (12 tile + 2 nu + pair); row r at + 2 r
  auto pass_a = [&](int sb) {
    if (!arole) return;
    float* S = wsm + 2 * W4_STAGE + sb * W4_SCRATCH;
    f32x2 o[6];
    wino4_bt<f32x2>(rawv, o);
    const int base = (12 * a_tile + a_q) * 12 + 2 * a_row;
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<f32x2*>(S + base + nu * 24) = o[nu];
  };
  auto pass_b = [&](int sb, int buf) {
    if (!arole) return;
    const float* S = wsm + 2 * W4_STAGE + sb * W4_SCRATCH + tid * 12;  // thread = (tile, nu, pair) = same split
    float* Vs = wsm + buf * W4_STAGE + W4U_STAGE;
    const float4 x0 = *reinterpret_cast<const float4*>(S);
    const float4 x1 = *reinterpret_cast<const float4*>(S + 4);
    const float4 x2 = *reinterpret_cast<const float4*>(S + 8);
    const f32x2 d[6] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}, {x2.x, x2.y}, {x2.z, x2.w}};
    f32x2 o[6];
    wino4_bt<f32x2>(d, o);
    const int nu = a_row;  // (tid % 12) >> 1 names nu in this pass
    const int base = nu * W4V_F + a_tile * W4K + 2 * a_q;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) *reinterpret_cast<f32x2*>(Vs + xi * 6 * W4V_F + base) = o[xi];
  };

  f32x16 acc[9];
#pragma unroll
  for (int f = 0; f < 9; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  // ---- prologue: stage 0 fully staged, stage 1's patch rows in registers
  if (nst > 0) {
    load_v(kt_begin);
    load_u(kt_begin);
    pass_a(0);
    store_u(0);
    if (nst > 1) load_v(kt_begin + 1);
  }
  __syncthreads();
  if (nst > 0) pass_b(0, 0);
  if (nst > 1) pass_a(1);
  __syncthreads();
  if (nst > 2) load_v(kt_begin + 2);

  // fragment rows: U row = c2*32 + l31, V row = l31; lane half h holds k = 2h, 2h+1
  const int f0 = (3 * gI) * 6 + 3 * gJ;
  const int fragA = f0 * W4U_F + (c2 * 32 + l31) * W4K + 2 * h;
  const int fragB = f0 * W4V_F + l31 * W4K + 2 * h;
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    // invariant here: stage s complete in buffers `buf`; scratch[(s+1)&1] holds pass-A rows of stage s+1;
    // rawv holds the patch rows of stage s+2
    if (s + 1 < nst) {
      load_u(kt_begin + s + 1);
      pass_b((s + 1) & 1, buf ^ 1);  // V buffer buf^1 was last read in iteration s-1
    }
    const float* Us = wsm + buf * W4_STAGE;
    const float* Vs = Us + W4U_STAGE;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f32x2 fa[3], fb[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        fa[j] = *reinterpret_cast<const f32x2*>(Us + (i * 6 + j) * W4U_F + fragA);
        fb[j] = *reinterpret_cast<const f32x2*>(Vs + (i * 6 + j) * W4V_F + fragB);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[i * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].x, fb[j].x, acc[i * 3 + j], 0, 0, 0);
        acc[i * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].y, fb[j].y, acc[i * 3 + j], 0, 0, 0);
      }
    }
    if (s + 1 < nst) store_u(buf ^ 1);
    if (s + 2 < nst) pass_a(s & 1);  // scratch[s&1] (stage s's rows) was consumed by pass B in iteration s-1
    __syncthreads();
    if (s + 3 < nst) load_v(kt_begin + s + 3);
  }

  // ---- output transform + exchange.  Wave (c2, I, J) reduces its 3x3 frequency block to a partial 4x4
  // output; for each register group g the waves of frequency block g collect the other three partials
  // through LDS and run the fused epilogue, so all eight waves share the stores.
  // Lane: tile = l31; couts c2*32 + 8g + 4h + (0..3) in accumulator registers 4g .. 4g+3.
  const int grp = gI + 2 * gJ;
  float* xch = wsm;  // [c2][frequency block 0..3][64 values][64 lanes] floats = 128 KB (stage buffers are dead)
  const int t = t0 + l31;
  const bool tvalid = t < p.ntiles;
  const int tt = tvalid ? t : 0;
  const int ob = tt / tiles_img;
  const int orr = tt - ob * tiles_img;
  const int oth = orr / p.tiles_w;
  const int otw = orr - oth * p.tiles_w;
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {  // rolled: each pass consumes registers 0..3, then the accumulators rotate by 4
    float y[16][4];  // [i*4 + j][r]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float z[3][4];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        wino4_at_part(gJ, acc[i * 3 + 0][r], acc[i * 3 + 1][r], acc[i * 3 + 2][r], z[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float col[4];
        wino4_at_part(gI, z[0][j], z[1][j], z[2][j], col);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i * 4 + j][r] = col[i];
      }
    }
    {  // every wave parks its partial in LDS (slot = its frequency block); the leader's rolled pixel loop
       // then needs no register indexing and keeps one pixel's epilogue live at a time
      float* dstx = xch + ((c2 * 4 + grp) * 64) * 64 + lane;
#pragma unroll
      for (int v = 0; v < 16; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) dstx[(v * 4 + r) * 64] = y[v][r];
    }
    __syncthreads();
    if (grp == g && tvalid) {
      const int co = cb * W4C + c2 * 32 + 8 * g + 4 * h;
      if (co < a.cout_s) {
        const float* srcx = xch + (c2 * 4 * 64) * 64 + lane;
#pragma unroll 1
        for (int v = 0; v < 16; ++v) {
          const int oh = 4 * oth + (v >> 2), ow = 4 * otw + (v & 3);
          if (oh >= a.hout || ow >= a.wout) continue;
          float o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* q = srcx + (v * 4 + r) * 64;
            o4[r] = ((q[0] + q[64 * 64]) + q[2 * 64 * 64]) + q[3 * 64 * 64];
          }
          const int n = (ob * a.hout + oh) * a.wout + ow;
          const float4 vv = make_float4(o4[0], o4[1], o4[2], o4[3]);
          if (a.splitk > 1)
            *reinterpret_cast<float4*>(a.workspace + ((int64_t)blockIdx.y * p.npix + n) * a.cout_s + co) = vv;
          else
            epilogue_store(a, n, co, vv);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 9; ++f)
      acc[f] = __builtin_shufflevector(acc[f], acc[f], 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3);
  }
}

}  // namespace

// bf16x3 token GEMMs / 1x1 convolutions: how many 256-cout tiles go to the 256 x 256 kernel (conv_gemm_x3_big_kernel; the rest of
// the output channels, if any, to the 128 x 128 kernel in a second launch) and which split-K keeps its ONE workgroup per CU in
// whole rounds.  Measured on 16384 tokens (tools/x3_probe.sh): 768 -> 3072 (768 big tiles = 3 rounds) 389 vs 455 us, 768 -> 768
// (192 tiles, 0.75 of a round) 117 vs 125, 768 -> 2304 (576 tiles = 2.25 rounds) a tie -- hence 8 of its 9 cout tiles big: 311 vs 338;
// 3072 -> 768 387 vs 444 (and no split-K combine).
// AZ_X3_BIG = 0 / 1: never / every eligible launch whole (A/B measurements; read per call, honoured only with AZ_DEBUG_AB set).
static bool x3_big_taps(const AzConvArgs* a) { return !(a->ksize == 1 && a->stride == 1 && a->pad == 0); }
static bool x3_big_eligible(const AzConvArgs* a, int64_t npix, int kstep = GBK) {  // kstep: 16 (bf16x3) / 64 (half-precision operands)
  const int64_t spix = (int64_t)a->batch * a->h0 * a->w0;
  if (a->src1 && !(a->up1 == 0 && a->h1 == a->hin && a->w1 == a->win && a->c1s % kstep == 0 && spix * a->c1s * 4 < (1ll << 31))) return false;
  if (x3_big_taps(a) && a->pad_mode != 0) return false;  // (filters with taps: zero padding only)
  return a->up0 == 0 && !a->aniso && a->depth == 0 && !a->dst_nchw && a->h0 == a->hin && a->w0 == a->win && a->c0s % kstep == 0 &&
         a->c0s + a->c1s >= 64 && spix * a->c0s * 4 < (1ll << 31);
}
static double x3_round_eff(int64_t wgs) { return (double)wgs / (double)(((wgs + 255) / 256) * 256); }
// -> number of 256-cout tiles for the big kernel (0: none); *splitk = the split-K it wants (1 unless the K loop is deep)
// *ct (bf16x3 kernel only) = the tile's couts: 256, or 192 where that makes whole rounds (768 = 4 x 192 and 2304 = 12 x 192 against
// 3 and 9 x 256 on 64 pixel tiles: 256 and 768 workgroups instead of 192 and 576); not for the SwiGLU / q-k preparation epilogues,
// whose lane maps assume 32 channel quads per pixel row.  AZ_X3_BIG = 3 forces it (A/B).
static int x3_big_plan(const AzConvArgs* a, int64_t npix, int* splitk, int kstep = GBK, int* ct = nullptr) {
  *splitk = 1;
  if (ct) *ct = GB;
  if (!x3_big_eligible(a, npix, kstep)) return 0;
  const char* force = az_ab_env("AZ_X3_BIG");  // (A/B override: only under the explicit debug switch AZ_DEBUG_AB)
  const int all = (a->cout_s + GB - 1) / GB;
  const bool ok192 = ct != nullptr && kstep == GBK && a->act <= 3 && !x3_big_taps(a);
  const int all192 = (a->cout_s + 191) / 192;
  if (force && force[0]) {
    if (force[0] == '1' && force[1] == ',') *splitk = atoi(force + 2);  // "1,S": every eligible launch on big tiles with split-K S (A/B)
    if (force[0] == '3' && ok192) {
      *ct = 192;
      return all192;
    }
    return force[0] == '1' || force[0] == '3' ? all : 0;
  }
  const int64_t tn = (npix + GB - 1) / GB;
  if (a->cout_s < 192) return 0;
  if (ok192 && all192 * tn >= 176) {
    const double e192 = x3_round_eff(all192 * tn) * a->cout_s / (all192 * 192.0), e256 = x3_round_eff(all * tn) * a->cout_s / (all * 256.0);
    if (e192 >= 0.9 && e192 > e256 + 0.08) {
      *ct = 192;
      return all192;
    }
  }
  if (all * tn < 176) {  // (128 tiles = half a round: 74 vs 62 us on 16384 x 512 -> 512; 192 tiles win)
    // ... unless the K loop is deep enough to split in two: 9216 x 2048 -> 768 (108 tiles) 168 us as 216 half-K tiles against 182 on
    // 128 x 128 tiles and 210 unsplit; 768-channel K loops lose that way (84 vs 79 us)
    if (all * tn * 2 >= 176 && (int64_t)a->ksize * a->ksize * (a->c0s + a->c1s) >= 1536) {
      *splitk = 2;
      return all;
    }
    return 0;
  }
  if (x3_round_eff(all * tn) >= 0.85) return all;
  // (no split-K, however deep the K loop: 16384 x 3072 -> 768 as 192 unsplit tiles 387 us, split 4 ways into whole rounds 444)
  for (int nb = a->cout_s / GB; nb >= 2; --nb)  // whole rounds of big tiles, the remaining output channels on the 128 x 128 kernel
    if (x3_round_eff(nb * tn) >= 0.95 && 2 * nb >= all) return nb;
  return all;  // (a partial round still beats the small tile: 192 tiles 117 vs 125 us)
}

__attribute__((visibility("hidden"))) int azi_winograd_x3_launch(const WinoP& p, unsigned splitk, hipStream_t st, bool h2);  // wino_x3.hip

extern "C" {

int az_conv2d_suggest_splitk(int64_t npix, int32_t cout_s, int32_t cin_s, int32_t ksize);
/* Split-K of az_conv2d_x3_f32 for a filled descriptor (everything but splitk / workspace): the 256 x 256 kernel's own choice where
 * it takes the launch (x3_big_plan), az_conv2d_suggest_splitk's otherwise. */
int az_conv2d_x3_suggest_splitk(const AzConvArgs* a) {
  if (!a) return 1;
  const int64_t npix = (int64_t)a->batch * a->hout * a->wout;
  int sk = 1, ct = GB;
  if (x3_big_plan(a, npix, &sk, GBK, &ct) > 0) return sk;  // (the same call as the launch: fill act / the q-k fields before asking)
  return az_conv2d_suggest_splitk(npix, a->cout_s, a->c0s + a->c1s, a->ksize);
}
int az_conv2d_suggest_splitk(int64_t npix, int32_t cout_s, int32_t cin_s, int32_t ksize) {
  const int64_t tiles = ((npix + BN - 1) / BN) * ((cout_s + BM - 1) / BM);
  const int64_t nk = (int64_t)ksize * ksize * ((cin_s + BK - 1) / BK);  // (two-source convs: within +1 per tap)
  // Fill 256 CUs x 2 resident blocks; keep >= 8 K-tiles per split so the slab traffic
  // (2 x 4 B x outputs per split) stays small next to the operand traffic.
  int64_t want = (512 + tiles - 1) / tiles;
  int64_t maxs = nk / 8;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > 32) want = 32;
  if (want < 1) want = 1;
  if (tiles >= 384) {
    // Enough tiles to fill the chip; split only to repair wave quantisation.  512 resident blocks: e.g. 768
    // tiles run as 2 rounds at 75 % occupancy, 2 x 768 = 1536 as exactly 3.  Worth the slab traffic only for
    // deep K (>= 48 K-tiles: the token GEMMs 3072 -> 768 of DiT-B, 2048 -> 768 of JiT-B).
    want = 1;
    if (nk >= 48 && tiles < 4096) {
      auto eff = [&](int64_t s) {
        const int64_t blocks = tiles * s;
        return (double)blocks / (double)(((blocks + 511) / 512) * 512);
      };
      if (eff(2) >= eff(1) + 0.15) want = 2;
    }
  }
  return (int)want;
}

// Split-K combine + fused epilogue; with AzConvArgs.gn_quads also the GroupNorm moments of the output, one partial per
// (image, chunk of ceil(hw / gn_chunks) pixels, channel quad).
static int launch_splitk_reduce(const ConvP& cp, hipStream_t st, int io = 0) {  // io: 0 fp32, 1 bf16, 2 f16 destination / residual
  const AzConvArgs& a = cp.a;
  if (a.gn_quads != nullptr) {
    AZ_REQUIRE(a.gn_chunks >= 1 && !a.dst_nchw && a.act != 4, AZ_E_UNSUPPORTED);
    const int hw = a.hout * a.wout;
    const int cpix = (hw + a.gn_chunks - 1) / a.gn_chunks;
    AZ_REQUIRE((int64_t)cpix * (a.gn_chunks - 1) < hw, AZ_E_SHAPE);  // no empty chunk
    const dim3 grid((unsigned)a.gn_chunks, (unsigned)a.batch);
    if (io == 1) hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel<1>, grid, dim3(256), 0, st, cp, cpix);
    else if (io == 2) hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel<2>, grid, dim3(256), 0, st, cp, cpix);
    else hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel<0>, grid, dim3(256), 0, st, cp, cpix);
  } else {
    const int grid = az_stream_grid((int64_t)cp.npix * (a.cout_s / 4), 256);
    if (io == 1) hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3(grid), dim3(256), 0, st, cp);
    else if (io == 2) hipLaunchKernelGGL(conv_splitk_reduce_kernel<2>, dim3(grid), dim3(256), 0, st, cp);
    else hipLaunchKernelGGL(conv_splitk_reduce_kernel<0>, dim3(grid), dim3(256), 0, st, cp);
  }
  return az_launch_status();
}

static int conv2d_direct(const AzConvArgs* a, az_stream_t stream, int half /* 0 fp32, 1 bf16, 2 f16 operands, 3 fp32 as 3 x bf16, 4 fp32 as 2 x f16 */);

int az_conv2d_f32(const AzConvArgs* a, az_stream_t stream) { return conv2d_direct(a, stream, 0); }

/* Image stem (conv_stem_kernel): src0 = (batch, c0s, hin, win) PLANAR fp32 with c0s in 1..4 channels, weight =
 * (3, 3, c0s, cout_s) floats (tap-major, output channels contiguous), 3x3 / stride 1 / pad 1 (zero or circular), optional
 * bias, act 0 / 1, NHWC destination, optional gn_quads with gn_chunks = ceil(hin / 8) * ceil(win / 32) partials per image. */
int az_conv2d_stem_f32(const AzConvArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->src0 && a->weight && a->dst, AZ_E_NULL);
  AZ_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1 && !a->aniso && !a->src1 && a->c1s == 0 && a->up0 == 0 && !a->gate &&
                 !a->res && !a->dst_nchw && !a->in_affine && a->depth == 0 && a->splitk <= 1 && (a->act == 0 || a->act == 1),
             AZ_E_UNSUPPORTED);
  AZ_REQUIRE(a->c0s >= 1 && a->c0s <= 4 && a->cout_s > 0 && a->cout_s % 4 == 0 && a->batch > 0 && a->hin > 0 && a->win > 0 &&
                 a->hout == a->hin && a->wout == a->win && a->h0 == a->hin && a->w0 == a->win,
             AZ_E_SHAPE);
  AZ_REQUIRE((int64_t)a->batch * a->hin * a->win * a->cout_s < (1ll << 40), AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(a->weight) && AZ_ALIGNED16(a->bias) && AZ_ALIGNED16(a->dst) && AZ_ALIGNED16(a->gn_quads), AZ_E_ALIGN);
  const int tiles_h = (a->hin + ST_TH - 1) / ST_TH, tiles_w = (a->win + ST_TW - 1) / ST_TW;
  const int tiles_img = tiles_h * tiles_w;
  AZ_REQUIRE((int64_t)a->batch * tiles_img < (1ll << 31), AZ_E_SHAPE);
  if (a->gn_quads) AZ_REQUIRE(a->gn_chunks == tiles_img, AZ_E_SHAPE);
  ConvP p{};
  p.a = *a;
  const dim3 grid((unsigned)(a->batch * tiles_img));
  hipStream_t st = az_s(stream);
  switch (a->c0s) {
    case 1: hipLaunchKernelGGL(conv_stem_kernel<1>, grid, dim3(256), 0, st, p, tiles_w, tiles_img); break;
    case 2: hipLaunchKernelGGL(conv_stem_kernel<2>, grid, dim3(256), 0, st, p, tiles_w, tiles_img); break;
    case 3: hipLaunchKernelGGL(conv_stem_kernel<3>, grid, dim3(256), 0, st, p, tiles_w, tiles_img); break;
    default: hipLaunchKernelGGL(conv_stem_kernel<4>, grid, dim3(256), 0, st, p, tiles_w, tiles_img); break;
  }
  return az_launch_status();
}

/* Same operation with bf16 / f16 MFMA operands and fp32 accumulation: `weight` is the 2-byte packing of
 * az_pack_conv_weight_half_f32, activations and outputs stay fp32 (rounded to the operand type inside the kernel).
 * For modules cast to half precision (azula/denoise.py:314-320); error ~2^-9 (bf16) / 2^-12 (f16) per product.   */
int az_conv2d_bf16_f32(const AzConvArgs* a, az_stream_t stream) { return conv2d_direct(a, stream, 1); }
int az_conv2d_f16_f32(const AzConvArgs* a, az_stream_t stream) { return conv2d_direct(a, stream, 2); }
/* fp32 operands evaluated as 3 x bf16 pieces / 6 partial products (see conv_igemm_x3_kernel): `weight` is the packing
 * of az_pack_conv_weight_x3_f32; fp32-level accuracy at 0.375 x the matrix-pipe time of the fp32 MFMA.              */
int az_conv2d_x3_f32(const AzConvArgs* a, az_stream_t stream) { return conv2d_direct(a, stream, 3); }
/* fp32 operands as 2 x f16 pieces / 3 partial products (include/azula_amd.h: "f16x2"): `weight` = az_pack_conv_weight_f16x2_f32
 * output, `w_scale` the scale given to it; the x3 kernels' tiles, plans and epilogues.                                   */
int az_conv2d_f16x2_f32(const AzConvArgs* a, az_stream_t stream) { return conv2d_direct(a, stream, 4); }

static bool conv_pow2(float v) {  // a finite positive power of two
  int e;
  return v > 0.f && v < 3.0e38f && frexpf(v, &e) == 0.5f;
}

static void launch_igemm_half(const ConvP& p, bool f16, bool srch, dim3 grid, hipStream_t st) {
  if (f16 && srch) hipLaunchKernelGGL((conv_igemm_half_kernel<true, true>), grid, dim3(256), 0, st, p);
  else if (f16) hipLaunchKernelGGL((conv_igemm_half_kernel<true, false>), grid, dim3(256), 0, st, p);
  else if (srch) hipLaunchKernelGGL((conv_igemm_half_kernel<false, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv_igemm_half_kernel<false, false>), grid, dim3(256), 0, st, p);
}

static int conv2d_direct(const AzConvArgs* a, az_stream_t stream, int half) {
  AZ_REQUIRE(!a || !a->in_affine, AZ_E_UNSUPPORTED);  // (the Winograd kernel's gather only)
  if (a && a->depth != 0)  // one depth tap of a 3-D convolution (both sources hold `batch` planes)
    AZ_REQUIRE(a->depth > 0 && a->batch % a->depth == 0 && a->depth_shift > -a->depth && a->depth_shift < a->depth &&
                   a->cout_s != 4 && (!a->gate || a->gate_bstride == 0) && (a->depth_wrap == 0 || a->depth_wrap == 1),
               AZ_E_UNSUPPORTED);
  AZ_REQUIRE(a && a->src0 && a->weight && a->dst, AZ_E_NULL);
  AZ_REQUIRE(a->batch > 0 && a->hin > 0 && a->win > 0 && a->hout > 0 && a->wout > 0, AZ_E_SHAPE);
  if (a->depth > 0) {  // 32-bit offsets from the descriptor's first plane: a tile's images + the planes a tap may reach back
    const int64_t span = (int64_t)BN / ((int64_t)a->hout * a->wout) + 2 + a->depth;  // (behind the shape checks: hout * wout > 0)
    AZ_REQUIRE(span * a->h0 * a->w0 * a->c0s * 4 < (1ll << 31) && span * a->h1 * a->w1 * a->c1s * 4 < (1ll << 31), AZ_E_SHAPE);
  }
  AZ_REQUIRE(a->c0s > 0 && a->c0s % 4 == 0 && a->c1s % 4 == 0 && a->cout_s > 0 && a->cout_s % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE((a->c1s == 0) == (a->src1 == nullptr), AZ_E_SHAPE);
  // half-precision activations in HBM (sources / destination + residual in the entry's 2-byte type): the bf16 / f16 entries only
  AZ_REQUIRE((a->src_dtype == 0 || a->src_dtype == 1) && (a->dst_dtype == 0 || a->dst_dtype == 1), AZ_E_SHAPE);
  AZ_REQUIRE((a->src_dtype == 0 && a->dst_dtype == 0) || half == 1 || half == 2, AZ_E_UNSUPPORTED);
  if (a->src_dtype) AZ_REQUIRE(a->c0s % 8 == 0 && a->c1s % 8 == 0, AZ_E_SHAPE);  // (16-byte loads of 8 values)
  if (a->dst_dtype) AZ_REQUIRE(!a->dst_nchw, AZ_E_UNSUPPORTED);                   // (the planar destination is the network's fp32 output)
  AZ_REQUIRE(a->act >= 0 && a->act <= 6, AZ_E_UNSUPPORTED);
  if (a->act == 6) AZ_REQUIRE(a->res && !a->gate && !a->dst_nchw && !a->gn_quads, AZ_E_UNSUPPORTED);  // SiLU of the sum with the residual
  if (a->act == 4) AZ_REQUIRE(!a->gate && !a->res && !a->dst_nchw && !a->gn_quads && a->cout_s % 8 == 0, AZ_E_UNSUPPORTED);  // SwiGLU epilogue
  if (a->act == 5) {  // q / k preparation of a fused qkv projection (epilogue_batch_qk)
    AZ_REQUIRE(!a->gate && !a->res && !a->dst_nchw && !a->gn_quads && a->splitk <= 1 && a->depth == 0, AZ_E_UNSUPPORTED);
    AZ_REQUIRE((a->qk_head_dim == 32 || a->qk_head_dim == 64 || a->qk_head_dim == 128) && a->qk_heads > 0 &&
                   a->cout_s == 3 * a->qk_heads * a->qk_head_dim && a->qk_tokens > 0 && a->hout * a->wout == a->qk_tokens,
               AZ_E_SHAPE);
    AZ_REQUIRE((a->qk_rope_cos == nullptr) == (a->qk_rope_sin == nullptr), AZ_E_NULL);
    AZ_REQUIRE(AZ_ALIGNED16(a->qk_q_weight) && AZ_ALIGNED16(a->qk_k_weight) && AZ_ALIGNED16(a->qk_rope_cos) && AZ_ALIGNED16(a->qk_rope_sin),
               AZ_E_ALIGN);
  }
  AZ_REQUIRE(a->ksize >= 1 && a->ksize <= 7 && a->stride >= 1 && a->pad >= 0 && (!a->aniso || a->stride_w >= 1), AZ_E_SHAPE);
  AZ_REQUIRE((a->hin + 2 * a->pad - a->ksize) / a->stride + 1 == a->hout &&
                 (a->win + 2 * a->pad - a->ksize) / (a->aniso ? a->stride_w : a->stride) + 1 == a->wout,
             AZ_E_SHAPE);
  AZ_REQUIRE(!a->aniso || (a->up0_w >= 0 && a->up0_w <= 4 && a->up1_w >= 0 && a->up1_w <= 4), AZ_E_SHAPE);  // (shift amounts)
  AZ_REQUIRE(a->up0 >= 0 && a->up0 <= 4 && ((a->hin + (1 << a->up0) - 1) >> a->up0) <= a->h0 &&
                 ((a->win + (1 << az_upw(a, a->up0, a->up0_w)) - 1) >> az_upw(a, a->up0, a->up0_w)) <= a->w0,
             AZ_E_SHAPE);
  if (a->src1)
    AZ_REQUIRE(a->up1 >= 0 && a->up1 <= 4 && ((a->hin + (1 << a->up1) - 1) >> a->up1) <= a->h1 &&
                 ((a->win + (1 << az_upw(a, a->up1, a->up1_w)) - 1) >> az_upw(a, a->up1, a->up1_w)) <= a->w1,
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(a->src0) && AZ_ALIGNED16(a->src1) && AZ_ALIGNED16(a->weight) && AZ_ALIGNED16(a->bias) &&
                 AZ_ALIGNED16(a->gate) && AZ_ALIGNED16(a->res) && AZ_ALIGNED16(a->workspace),
             AZ_E_ALIGN);
  if (!a->dst_nchw) AZ_REQUIRE(AZ_ALIGNED16(a->dst), AZ_E_ALIGN);
  if (a->dst_nchw) AZ_REQUIRE(a->dst_c > 0 && a->dst_c <= a->cout_s, AZ_E_SHAPE);
  if (a->gate) AZ_REQUIRE(a->gate_bstride % 4 == 0, AZ_E_ALIGN);
  if (a->res && a->res_up) AZ_REQUIRE(((a->hout + 1) >> 1) <= a->hres && ((a->wout + 1) >> 1) <= a->wres, AZ_E_SHAPE);
  AZ_REQUIRE(a->splitk >= 1 && (a->splitk == 1 || a->workspace), AZ_E_SHAPE);
  const int64_t npix64 = (int64_t)a->batch * a->hout * a->wout;
  AZ_REQUIRE(npix64 < (1ll << 31), AZ_E_SHAPE);

  const bool x3 = half == 3 || half == 4;  // fp32 operands as pieces: the same kernels, plans and packed layout (three 2-byte planes)
  if (half == 4) AZ_REQUIRE(conv_pow2(a->w_scale), AZ_E_SHAPE);
  ConvP p;
  p.a = *a;
  p.npix = (int)npix64;
  p.cin_s = a->c0s + a->c1s;
  p.out_scale = half == 4 ? 1.f / a->w_scale : 1.f;
  if (half == 4) AZ_REQUIRE(AZ_ALIGNED16(a->in_absmax0) && AZ_ALIGNED16(a->in_absmax1) && (a->in_absmax0 || !a->in_absmax1), AZ_E_ALIGN);
  // fp32 direct kernel: K tile 16 with three workgroups per CU when the workgroup count then fills the chip evenly
  // (see conv_igemm_kernel); AZ_IGEMM_K16 = 0 / 1 forces the choice (A/B measurements)
  bool k16 = false;
  if (!half) {
    const char* force = az_ab_env("AZ_IGEMM_K16");  // read per call: tests flip it between plans (only under AZ_DEBUG_AB)
    int sk = a->splitk;
    const int64_t nk32 = (int64_t)a->ksize * a->ksize * ((a->c0s + BK - 1) / BK + (a->c1s + BK - 1) / BK);
    if (sk > nk32) sk = (int)nk32;
    const int64_t wgs = (int64_t)((a->cout_s + BM - 1) / BM) * ((npix64 + BN - 1) / BN) * sk;
    const int64_t n = (wgs + 255) / 256;  // workgroups on the fullest CU
    auto cost = [](int64_t n, int slots) {  // in tile times: a lone workgroup keeps the matrix pipe ~65 % busy
      const int64_t r = n % slots;
      return (double)(n - r) + (r == 0 ? 0.0 : (r == 1 ? 1.0 / 0.65 : (double)r));
    };
    // measured (C3 token GEMMs, round 3): per tile the K-16 instantiation now equals the K-32 one (768 -> 3072: 676 vs 670 us;
    // its epilogue no longer spills), so it is taken whenever three slots per CU quantise better (768 -> 768: 184 vs 221 us,
    // 768 -> 2304: 519 vs 548 us)
    k16 = n >= 3 && cost(n, 3) * 1.01 + 0.05 < cost(n, 2);
    // one tap, whole 32-channel K tiles: the K-32 kernel runs its hand-scheduled K loop (igemm_kloop.inc), which keeps the
    // matrix pipe busy from ONE workgroup per CU as well -- it beats the K-16 instantiation on the badly quantised shapes too
    // (768 -> 768: 178 vs 184 us, 768 -> 2304: 492 vs 519 us).  AZ_IGEMM_ASM=0: the C++ K loop everywhere (A/B measurements)
    const char* asm_env = az_ab_env("AZ_IGEMM_ASM");
    const bool asm_ok = a->ksize == 1 && a->c0s % BK == 0 && a->c1s % BK == 0 && !(asm_env && asm_env[0] == '0');
    // several taps: the stream's taps variant (pixel-row offsets linear in the tap + a validity mask per row); it needs ONE
    // source, no upsampling, zero padding, isotropic strides, at most 32 taps, and pays from 16 stages per split-K slice on
    // (measured: 256^2 stride-2 256 -> 256: 672 -> 633 us, 128^2: 343 -> 328; 9-stage slices of an 8 x 8 map: 63 -> 66).
    // AZ_IGEMM_ASM=1: one-tap stream only (A/B)
    const bool taps_ok = a->ksize > 1 && a->ksize * a->ksize <= 32 && !a->src1 && a->c0s % BK == 0 && a->up0 == 0 &&
                         a->pad_mode == 0 && !a->aniso && !(asm_env && asm_env[0]) && nk32 / (sk < 1 ? 1 : sk) >= 16 &&
                         (int64_t)a->h0 * a->w0 * a->c0s * 4 * a->batch < (1ll << 31);
    if (asm_ok || taps_ok) k16 = false;
    if (force) k16 = force[0] == '1';
    p.asm_loop = k16 ? 0 : (asm_ok ? 1 : (taps_ok ? 2 : 0));
  } else {
    p.asm_loop = 0;
  }
  int big_sk = 1;
  // 256-cout tiles of the 256 x 256 kernel (bf16x3: K steps of 16 channels; half-precision operands: 64)
  int big_ct = GB;
  int nbig = x3 ? x3_big_plan(a, npix64, &big_sk, GBK, &big_ct) : (half == 1 || half == 2) ? x3_big_plan(a, npix64, &big_sk, HGK) : 0;
  if (nbig > 0 && nbig * big_ct < a->cout_s && a->splitk > 1) nbig = 0;  // (a remainder launch would need the same slabs: one kernel then)
  const bool big = nbig > 0;
  const int bk = big ? (x3 ? GBK : HGK) : x3 ? XBK : (half ? HBK : (k16 ? 16 : BK));
  p.nkc0 = (a->c0s + bk - 1) / bk;
  p.nkc1 = (a->c1s + bk - 1) / bk;
  p.nk = a->ksize * a->ksize * (p.nkc0 + p.nkc1);
  // 32-bit relative byte offsets inside the kernel: a pixel tile spans at most
  // ceil(BN / (hout*wout)) + 1 samples of either source.
  {
    const int64_t span = (BN + (int64_t)a->hout * a->wout - 1) / ((int64_t)a->hout * a->wout) + 1;
    AZ_REQUIRE(span * a->h0 * a->w0 * a->c0s * 4 < (1ll << 31), AZ_E_SHAPE);
    AZ_REQUIRE(span * a->h1 * a->w1 * a->c1s * 4 < (1ll << 31), AZ_E_SHAPE);
    AZ_REQUIRE((int64_t)a->cout_s * p.cin_s * 4 < (1ll << 31), AZ_E_SHAPE);
    AZ_REQUIRE((int64_t)a->ksize * a->ksize * a->cout_s * p.cin_s * (x3 ? 6 : 4) <= (1ll << 31), AZ_E_SHAPE);
  }
  hipStream_t st = az_s(stream);
  if (!half && a->cout_s == 4 && a->ksize == 3 && a->stride == 1 && a->pad == 1 && !a->src1 && a->up0 == 0 && !a->aniso &&
      a->c0s % HD_KC == 0 && a->h0 == a->hin && a->w0 == a->win) {
    // narrow output (image head): VALU kernel, no split-K
    p.a.splitk = 1;
    const int tiles_x = (a->wout + HD_T - 1) / HD_T, tiles_y = (a->hout + HD_T - 1) / HD_T;
    const dim3 grid((unsigned)(tiles_x * tiles_y * a->batch));
    const int nco = a->dst_nchw ? a->dst_c : 4;
    if (nco == 1) hipLaunchKernelGGL(conv_head_kernel<1>, grid, dim3(256), 0, st, p, tiles_x, tiles_y);
    else if (nco == 2) hipLaunchKernelGGL(conv_head_kernel<2>, grid, dim3(256), 0, st, p, tiles_x, tiles_y);
    else if (nco == 3) hipLaunchKernelGGL(conv_head_kernel<3>, grid, dim3(256), 0, st, p, tiles_x, tiles_y);
    else hipLaunchKernelGGL(conv_head_kernel<4>, grid, dim3(256), 0, st, p, tiles_x, tiles_y);
    return az_launch_status();
  }
  int splitk = a->splitk;
  if (splitk > p.nk) splitk = p.nk;
  p.kps = (p.nk + splitk - 1) / splitk;
  splitk = (p.nk + p.kps - 1) / p.kps;  // no empty splits
  p.a.splitk = splitk;
  // GroupNorm moments of the output: from the split-K combine only (the direct kernels' own epilogues do not produce them)
  AZ_REQUIRE(!a->gn_quads || splitk > 1, AZ_E_UNSUPPORTED);
  p.a.gn_quads = nullptr;
  p.tiles_m = (a->cout_s + BM - 1) / BM;
  p.tiles_n = (p.npix + BN - 1) / BN;
  p.m_tile0 = 0;
  if (big) {
    p.tiles_m = nbig;
    p.tiles_n = (p.npix + GB - 1) / GB;
  }
  int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  AZ_REQUIRE(nwg < (1ll << 31), AZ_E_SHAPE);
  if (big) {
    if (half == 3 && big_ct == 192) hipLaunchKernelGGL(conv_gemm_x3_big_kernel<3>, dim3((unsigned)nwg, (unsigned)splitk), dim3(512), 0, st, p);
    else if (half == 3 && x3_big_taps(a)) hipLaunchKernelGGL((conv_gemm_x3_big_kernel<4, true>), dim3((unsigned)nwg, (unsigned)splitk), dim3(512), 0, st, p);
    else if (half == 3) hipLaunchKernelGGL(conv_gemm_x3_big_kernel<4>, dim3((unsigned)nwg, (unsigned)splitk), dim3(512), 0, st, p);
    else if (half == 4 && big_ct == 192) hipLaunchKernelGGL((conv_gemm_x3_big_kernel<3, false, true>), dim3((unsigned)nwg, (unsigned)splitk), dim3(512), 0, st, p);
    else if (half == 4 && x3_big_taps(a)) hipLaunchKernelGGL((conv_gemm_x3_big_kernel<4, true, true>), dim3((unsigned)nwg, (unsigned)splitk), dim3(512), 0, st, p);
    else if (half == 4) hipLaunchKernelGGL((conv_gemm_x3_big_kernel<4, false, true>), dim3((unsigned)nwg, (unsigned)splitk), dim3(512), 0, st, p);
    else {
      const dim3 g((unsigned)nwg, (unsigned)splitk);
#define AZ_HBIG(F16, TAPS, SRCH) hipLaunchKernelGGL((conv_gemm_half_big_kernel<F16, TAPS, SRCH>), g, dim3(512), 0, st, p)
      switch ((half == 2 ? 4 : 0) + (x3_big_taps(a) ? 2 : 0) + (a->src_dtype ? 1 : 0)) {
        case 0: AZ_HBIG(false, false, false); break;
        case 1: AZ_HBIG(false, false, true); break;
        case 2: AZ_HBIG(false, true, false); break;
        case 3: AZ_HBIG(false, true, true); break;
        case 4: AZ_HBIG(true, false, false); break;
        case 5: AZ_HBIG(true, false, true); break;
        case 6: AZ_HBIG(true, true, false); break;
        default: AZ_HBIG(true, true, true); break;
      }
#undef AZ_HBIG
    }
    if (nbig * big_ct < a->cout_s) {  // the remaining output channels (256-cout tiles only): 128 x 128 tiles, K tiles of 32
      ConvP q = p;
      const int sbk = x3 ? XBK : HBK;
      q.nkc0 = (a->c0s + sbk - 1) / sbk;
      q.nkc1 = (a->c1s + sbk - 1) / sbk;
      q.nk = a->ksize * a->ksize * (q.nkc0 + q.nkc1);
      q.kps = (q.nk + splitk - 1) / splitk;
      AZ_REQUIRE((q.nk + q.kps - 1) / q.kps == splitk, AZ_E_SHAPE);  // (the same slabs as the big launch)
      q.m_tile0 = nbig * (GB / BM);
      q.tiles_m = (a->cout_s - nbig * GB + BM - 1) / BM;
      q.tiles_n = (p.npix + BN - 1) / BN;
      nwg = (int64_t)q.tiles_m * q.tiles_n;
      if (half == 3) hipLaunchKernelGGL(conv_igemm_x3_kernel<false>, dim3((unsigned)nwg, (unsigned)splitk), dim3(256), 0, st, q);
      else if (half == 4) hipLaunchKernelGGL(conv_igemm_x3_kernel<true>, dim3((unsigned)nwg, (unsigned)splitk), dim3(256), 0, st, q);
      else launch_igemm_half(q, half == 2, a->src_dtype != 0, dim3((unsigned)nwg, (unsigned)splitk), st);
    }
  } else
  if (half == 1 || half == 2)
    launch_igemm_half(p, half == 2, a->src_dtype != 0, dim3((unsigned)nwg, (unsigned)splitk), st);
  else if (half == 3)
    hipLaunchKernelGGL(conv_igemm_x3_kernel<false>, dim3((unsigned)nwg, (unsigned)splitk), dim3(256), 0, st, p);
  else if (half == 4)
    hipLaunchKernelGGL(conv_igemm_x3_kernel<true>, dim3((unsigned)nwg, (unsigned)splitk), dim3(256), 0, st, p);
  else if (k16)
    hipLaunchKernelGGL(conv_igemm_kernel<16>, dim3((unsigned)nwg, (unsigned)splitk), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(conv_igemm_kernel<32>, dim3((unsigned)nwg, (unsigned)splitk), dim3(256), 0, st, p);
  int rc = az_launch_status();
  if (rc != AZ_OK) return rc;
  if (splitk > 1) {
    p.a.gn_quads = a->gn_quads;
    rc = launch_splitk_reduce(p, st, a->dst_dtype ? half : 0);
  }
  return rc;
}

/* Winograd F(2x2,3x3) path: same AzConvArgs, but `weight` must be the filter transform packed by
 * the host (U = G g G^T laid out [chunk][cout block][16][32][8], see azula_amd/engine.py).  Only
 * ksize = 3, stride = 1, pad = 1. */
static int wino_prepare(const AzConvArgs* a, int wk, int64_t ustage_bytes, WinoP& p, int& splitk);

int az_conv2d_winograd_f32(const AzConvArgs* a, az_stream_t stream) {
  WinoP p;
  int splitk = 1;
  const int prc = wino_prepare(a, WK, (int64_t)WU_STAGE * 4, p, splitk);
  if (prc != AZ_OK) return prc;
  hipStream_t st = az_s(stream);
  const int64_t nwg = (int64_t)p.cblocks * p.tblocks;
  {
    static std::atomic<uint64_t> lds_c{0}, lds_s{0};  // (per kernel, one bit per device: common.h)
    hipError_t e = az_max_dynamic_lds((const void*)conv_winograd_kernel<false>, W_LDS_BYTES, lds_c);
    if (e != hipSuccess) return (int)e;
    e = az_max_dynamic_lds((const void*)conv_winograd_kernel<true>, W_LDS_BYTES + W_VOFF_BYTES, lds_s);
    if (e != hipSuccess) return (int)e;
  }
  // AZ_WINOGRAD_ASM=0: the C++ K loop (A/B measurements); default: the hand-scheduled stream
  const char* asm_env = az_ab_env("AZ_WINOGRAD_ASM");
  if ((asm_env && asm_env[0] == '0') || (a->in_affine && a->in_act != 0))  // (the stream has the plain affine only)
    hipLaunchKernelGGL(conv_winograd_kernel<false>, dim3((unsigned)nwg, (unsigned)splitk), dim3(512), W_LDS_BYTES, st, p);
  else
    hipLaunchKernelGGL(conv_winograd_kernel<true>, dim3((unsigned)nwg, (unsigned)splitk), dim3(512), W_LDS_BYTES + W_VOFF_BYTES, st, p);
  int rc = az_launch_status();
  if (rc != AZ_OK) return rc;
  if (splitk > 1) {
    ConvP cp;
    cp.a = p.a;
    cp.a.gn_quads = a->gn_quads;
    cp.npix = p.npix;
    rc = launch_splitk_reduce(cp, st);
  }
  return rc;
}

/* The same convolution with the 16 frequency GEMMs on the bf16 matrix pipe as exact 3 x bf16 splits (wino_x3.hip): `weight` =
 * az_winograd_pack_filter_x3_f32 output (16-channel steps); every other field as az_conv2d_winograd_f32. */
static int winograd_x3_entry(const AzConvArgs* a, az_stream_t stream, bool h2);
int az_conv2d_winograd_x3_f32(const AzConvArgs* a, az_stream_t stream) { return winograd_x3_entry(a, stream, false); }
/* The f16x2 form of the same kernel (include/azula_amd.h): `weight` = az_winograd_pack_filter_f16x2_f32 output, `w_scale` its scale. */
int az_conv2d_winograd_f16x2_f32(const AzConvArgs* a, az_stream_t stream) { return winograd_x3_entry(a, stream, true); }

static int winograd_x3_entry(const AzConvArgs* a, az_stream_t stream, bool h2) {
  WinoP p;
  int splitk = 1;
  const int prc = wino_prepare(a, 16, 16ll * WC * 16 * 3 * 2, p, splitk);
  if (prc != AZ_OK) return prc;
  AZ_REQUIRE(p.tiles_w >= 2, AZ_E_UNSUPPORTED);  // (the kernel stages <= 32 tile-row segments per 64-tile block: maps >= 3 pixels wide)
  if (h2) AZ_REQUIRE(conv_pow2(a->w_scale), AZ_E_SHAPE);
  p.out_scale = h2 ? 1.f / a->w_scale : 1.f;
  if (h2) AZ_REQUIRE(AZ_ALIGNED16(a->in_absmax0) && AZ_ALIGNED16(a->in_absmax1) && (a->in_absmax0 || !a->in_absmax1), AZ_E_ALIGN);
  hipStream_t st = az_s(stream);
  int rc = azi_winograd_x3_launch(p, (unsigned)splitk, st, h2);
  if (rc != AZ_OK) return rc;
  if (splitk > 1) {
    ConvP cp;
    cp.a = p.a;
    cp.a.gn_quads = a->gn_quads;
    cp.npix = p.npix;
    rc = launch_splitk_reduce(cp, st);
  }
  return rc;
}

/* Validation and launch geometry shared by the two Winograd entries; wk = input channels per K step (8 / 16). */
static int wino_prepare(const AzConvArgs* a, int wk, int64_t ustage_bytes, WinoP& p, int& splitk) {
  AZ_REQUIRE(a && a->src0 && a->weight && a->dst, AZ_E_NULL);
  AZ_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1 && !a->aniso && a->src_dtype == 0 && a->dst_dtype == 0, AZ_E_UNSUPPORTED);  // anisotropic, half-precision tensors: direct kernel
  AZ_REQUIRE(a->batch > 0 && a->hin > 0 && a->win > 0 && a->hout == a->hin && a->wout == a->win, AZ_E_SHAPE);
  AZ_REQUIRE(a->c0s > 0 && a->c0s % 4 == 0 && a->c1s % 4 == 0 && a->cout_s > 0 && a->cout_s % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE((a->c1s == 0) == (a->src1 == nullptr), AZ_E_SHAPE);
  AZ_REQUIRE((a->act >= 0 && a->act <= 4) || a->act == 6, AZ_E_UNSUPPORTED);
  if (a->act == 6) AZ_REQUIRE(a->res && !a->gate && !a->dst_nchw && !a->gn_quads, AZ_E_UNSUPPORTED);  // SiLU of the sum with the residual
  if (a->act == 4) AZ_REQUIRE(!a->gate && !a->res && !a->dst_nchw && !a->gn_quads && a->cout_s % 8 == 0, AZ_E_UNSUPPORTED);  // SwiGLU epilogue
  if (a->depth != 0) {  // one depth tap of a 3-D convolution
    AZ_REQUIRE(a->depth > 0 && a->batch % a->depth == 0 && a->depth_shift > -a->depth && a->depth_shift < a->depth &&
                   (!a->gate || a->gate_bstride == 0) && (a->depth_wrap == 0 || a->depth_wrap == 1),
               AZ_E_UNSUPPORTED);
    // 32-bit offsets from the descriptor's first plane: a tile block's images + the planes a tap may reach back
    const int64_t tiles_img = (int64_t)((a->hout + 1) / 2) * ((a->wout + 1) / 2);
    const int64_t span = (WT + tiles_img - 1) / tiles_img + 2 + a->depth;
    AZ_REQUIRE(span * a->h0 * a->w0 * a->c0s * 4 < (1ll << 31) && span * a->h1 * a->w1 * a->c1s * 4 < (1ll << 31), AZ_E_SHAPE);
  }
  if (a->in_affine)  // the normalisation apply pass inside the gather (the x3 kernel's patch masks live in output coordinates: it
                     // also takes a nearest-upsampled source)
    AZ_REQUIRE(!a->src1 && a->c0s % 8 == 0 && (a->up0 == 0 || wk == 16) && (a->in_act == 0 || a->in_act == 1) && AZ_ALIGNED16(a->in_affine) &&
                   (int64_t)a->batch * a->c0s * 8 < (1ll << 31),
               AZ_E_UNSUPPORTED);
  AZ_REQUIRE(!a->aniso || (a->up0_w >= 0 && a->up0_w <= 4 && a->up1_w >= 0 && a->up1_w <= 4), AZ_E_SHAPE);  // (shift amounts)
  AZ_REQUIRE(a->up0 >= 0 && a->up0 <= 4 && ((a->hin + (1 << a->up0) - 1) >> a->up0) <= a->h0 &&
                 ((a->win + (1 << az_upw(a, a->up0, a->up0_w)) - 1) >> az_upw(a, a->up0, a->up0_w)) <= a->w0,
             AZ_E_SHAPE);
  if (a->src1)
    AZ_REQUIRE(a->up1 >= 0 && a->up1 <= 4 && ((a->hin + (1 << a->up1) - 1) >> a->up1) <= a->h1 &&
                 ((a->win + (1 << az_upw(a, a->up1, a->up1_w)) - 1) >> az_upw(a, a->up1, a->up1_w)) <= a->w1,
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(a->src0) && AZ_ALIGNED16(a->src1) && AZ_ALIGNED16(a->weight) && AZ_ALIGNED16(a->bias) &&
                 AZ_ALIGNED16(a->gate) && AZ_ALIGNED16(a->res) && AZ_ALIGNED16(a->workspace),
             AZ_E_ALIGN);
  if (!a->dst_nchw) AZ_REQUIRE(AZ_ALIGNED16(a->dst), AZ_E_ALIGN);
  if (a->dst_nchw) AZ_REQUIRE(a->dst_c > 0 && a->dst_c <= a->cout_s, AZ_E_SHAPE);
  if (a->gate) AZ_REQUIRE(a->gate_bstride % 4 == 0, AZ_E_ALIGN);
  if (a->res && a->res_up) AZ_REQUIRE(((a->hout + 1) >> 1) <= a->hres && ((a->wout + 1) >> 1) <= a->wres, AZ_E_SHAPE);
  AZ_REQUIRE(a->splitk >= 1 && (a->splitk == 1 || a->workspace), AZ_E_SHAPE);
  const int64_t npix64 = (int64_t)a->batch * a->hout * a->wout;
  AZ_REQUIRE(npix64 < (1ll << 31), AZ_E_SHAPE);
  if (a->gn_quads && a->splitk == 1) {  // statistics of the output for a following GroupNorm: one tile block = 64 tiles of ONE image
    const int64_t tiles_img = (int64_t)((a->hout + 1) / 2) * ((a->wout + 1) / 2);
    AZ_REQUIRE(!a->dst_nchw && a->cout_s % WC == 0 && tiles_img % WT == 0 && a->hout % 2 == 0 &&
                   a->wout % 2 == 0 && a->gn_chunks == tiles_img / WT,  // whole 2 x 2 tiles only: 1024 values per partial
               AZ_E_UNSUPPORTED);
  }  // (split-K > 1: the moments come from the combine kernel, gn_chunks = pixel chunks per image)

  p.a = *a;
  p.npix = (int)npix64;
  p.tiles_h = (a->hout + 1) / 2;
  p.tiles_w = (a->wout + 1) / 2;
  p.ntiles = a->batch * p.tiles_h * p.tiles_w;
  p.nkc0 = (a->c0s + wk - 1) / wk;
  p.nkc1 = (a->c1s + wk - 1) / wk;
  p.nk = p.nkc0 + p.nkc1;
  {
    const int64_t tiles_img = (int64_t)p.tiles_h * p.tiles_w;
    const int64_t span = (WT + tiles_img - 1) / tiles_img + 1;
    AZ_REQUIRE(span * a->h0 * a->w0 * a->c0s * 4 < (1ll << 31), AZ_E_SHAPE);
    AZ_REQUIRE(span * a->h1 * a->w1 * a->c1s * 4 < (1ll << 31), AZ_E_SHAPE);
  }
  splitk = a->splitk;
  if (splitk > p.nk) splitk = p.nk;
  p.kps = (p.nk + splitk - 1) / splitk;
  splitk = (p.nk + p.kps - 1) / p.kps;
  p.a.splitk = splitk;
  AZ_REQUIRE(!a->gn_quads || (a->splitk == 1) == (splitk == 1), AZ_E_UNSUPPORTED);  // gn_chunks means tile blocks OR pixel chunks
  if (splitk > 1) p.a.gn_quads = nullptr;  // (the slabs carry no moments: the combine kernel produces them)
  p.cblocks = (a->cout_s + WC - 1) / WC;
  p.tblocks = (p.ntiles + WT - 1) / WT;
  {
    // workgroup order (see the kernel): rectangles of gt x gc = 32 workgroups when both grid sides divide, cout blocks fastest
    // otherwise (gt = 1, gc = cblocks).  AZ_WINO_RECT="gt,gc" overrides (A/B runs); "1,0" = cout fastest everywhere.
    int env_gt = 0, env_gc = 0;  // (read per call like the other A/B switches: no unsynchronised static)
    if (const char* e = az_ab_env("AZ_WINO_RECT")) sscanf(e, "%d,%d", &env_gt, &env_gc);
    // (the x3 kernel's filter chunks are 1.5 x the fp32 stream's -- 6 B per value: two cout blocks per XCD keep a layer's chunks in its
    //  4 MB L2 where four thrash it: 16 x 2 measured 1 - 2 % ahead of 8 x 4 on the 256- and 512-channel layers, tools/wx3_rect_ab.sh)
    int gt = env_gt > 0 ? env_gt : 8, gc = env_gt > 0 ? (env_gc > 0 ? env_gc : p.cblocks) : (wk == 16 ? 2 : 4);
    while (gc > 1 && p.cblocks % gc) gc >>= 1;
    if (env_gt <= 0) gt = 32 / gc;
    while (gt > 1 && p.tblocks % gt) gt >>= 1;
    if (p.cblocks % gc || p.tblocks % gt) gt = 1, gc = p.cblocks;
    p.gt = gt, p.gc = gc;
  }
  AZ_REQUIRE((int64_t)p.nk * p.cblocks * ustage_bytes <= (1ll << 31), AZ_E_SHAPE);
  return AZ_OK;
}

/* Split-K suggestion for the Winograd path (pure function of the shape). */
int az_conv2d_winograd_suggest_splitk(int64_t batch, int32_t hout, int32_t wout, int32_t cout_s, int32_t cin_s) {
  const int64_t tiles = batch * ((hout + 1) / 2) * ((wout + 1) / 2);
  const int64_t blocks = ((tiles + WT - 1) / WT) * ((cout_s + WC - 1) / WC);
  const int64_t nk = (cin_s + WK - 1) / WK;
  int64_t want = (256 + blocks - 1) / blocks;
  int64_t maxs = nk / 8;  // >= 8 stages per slice (4 x 8 x 8, 1024 -> 1024: 16 slices 37.7 us, 8 slices 51.0, 32 slices 48.6)
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > 16) want = 16;
  if (blocks >= 192) want = 1;
  return (int)(want < 1 ? 1 : want);
}

/* Winograd F(4x4,3x3) path (opt-in, see the kernel's header comment): same AzConvArgs, `weight` packed by
 * az_winograd4_pack_filter_f32.  Only ksize = 3, stride = 1, pad = 1. */
int az_conv2d_winograd4_f32(const AzConvArgs* a, az_stream_t stream) {
  AZ_REQUIRE(a && a->src0 && a->weight && a->dst, AZ_E_NULL);
  AZ_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1 && !a->aniso && a->src_dtype == 0 && a->dst_dtype == 0, AZ_E_UNSUPPORTED);  // anisotropic, half-precision tensors: direct kernel
  AZ_REQUIRE(a->batch > 0 && a->hin > 0 && a->win > 0 && a->hout == a->hin && a->wout == a->win, AZ_E_SHAPE);
  AZ_REQUIRE(a->c0s > 0 && a->c0s % 4 == 0 && a->c1s % 4 == 0 && a->cout_s > 0 && a->cout_s % 4 == 0, AZ_E_SHAPE);
  AZ_REQUIRE((a->c1s == 0) == (a->src1 == nullptr), AZ_E_SHAPE);
  AZ_REQUIRE((a->act >= 0 && a->act <= 4) || a->act == 6, AZ_E_UNSUPPORTED);
  if (a->act == 6) AZ_REQUIRE(a->res && !a->gate && !a->dst_nchw && !a->gn_quads, AZ_E_UNSUPPORTED);  // SiLU of the sum with the residual
  if (a->act == 4) AZ_REQUIRE(!a->gate && !a->res && !a->dst_nchw && !a->gn_quads && a->cout_s % 8 == 0, AZ_E_UNSUPPORTED);  // SwiGLU epilogue
  AZ_REQUIRE(!a->in_affine && a->depth == 0, AZ_E_UNSUPPORTED);
  AZ_REQUIRE(!a->aniso || (a->up0_w >= 0 && a->up0_w <= 4 && a->up1_w >= 0 && a->up1_w <= 4), AZ_E_SHAPE);  // (shift amounts)
  AZ_REQUIRE(a->up0 >= 0 && a->up0 <= 4 && ((a->hin + (1 << a->up0) - 1) >> a->up0) <= a->h0 &&
                 ((a->win + (1 << az_upw(a, a->up0, a->up0_w)) - 1) >> az_upw(a, a->up0, a->up0_w)) <= a->w0,
             AZ_E_SHAPE);
  if (a->src1)
    AZ_REQUIRE(a->up1 >= 0 && a->up1 <= 4 && ((a->hin + (1 << a->up1) - 1) >> a->up1) <= a->h1 &&
                 ((a->win + (1 << az_upw(a, a->up1, a->up1_w)) - 1) >> az_upw(a, a->up1, a->up1_w)) <= a->w1,
             AZ_E_SHAPE);
  AZ_REQUIRE(AZ_ALIGNED16(a->src0) && AZ_ALIGNED16(a->src1) && AZ_ALIGNED16(a->weight) && AZ_ALIGNED16(a->bias) &&
                 AZ_ALIGNED16(a->gate) && AZ_ALIGNED16(a->res) && AZ_ALIGNED16(a->workspace),
             AZ_E_ALIGN);
  if (!a->dst_nchw) AZ_REQUIRE(AZ_ALIGNED16(a->dst), AZ_E_ALIGN);
  if (a->dst_nchw) AZ_REQUIRE(a->dst_c > 0 && a->dst_c <= a->cout_s, AZ_E_SHAPE);
  if (a->gate) AZ_REQUIRE(a->gate_bstride % 4 == 0, AZ_E_ALIGN);
  if (a->res && a->res_up) AZ_REQUIRE(((a->hout + 1) >> 1) <= a->hres && ((a->wout + 1) >> 1) <= a->wres, AZ_E_SHAPE);
  AZ_REQUIRE(a->splitk >= 1 && (a->splitk == 1 || a->workspace), AZ_E_SHAPE);
  const int64_t npix64 = (int64_t)a->batch * a->hout * a->wout;
  AZ_REQUIRE(npix64 < (1ll << 31), AZ_E_SHAPE);

  Wino4P p;
  p.a = *a;
  p.npix = (int)npix64;
  p.tiles_h = (a->hout + 3) / 4;
  p.tiles_w = (a->wout + 3) / 4;
  p.ntiles = a->batch * p.tiles_h * p.tiles_w;
  p.nkc0 = a->c0s / W4K;
  p.nk = p.nkc0 + a->c1s / W4K;
  {
    const int64_t tiles_img = (int64_t)p.tiles_h * p.tiles_w;
    const int64_t span = (W4T + tiles_img - 1) / tiles_img + 1;
    AZ_REQUIRE(span * a->h0 * a->w0 * a->c0s * 4 < (1ll << 31), AZ_E_SHAPE);
    AZ_REQUIRE(span * a->h1 * a->w1 * a->c1s * 4 < (1ll << 31), AZ_E_SHAPE);
  }
  int splitk = a->splitk;
  if (splitk > p.nk) splitk = p.nk;
  p.kps = (p.nk + splitk - 1) / splitk;
  splitk = (p.nk + p.kps - 1) / p.kps;
  p.a.splitk = splitk;
  AZ_REQUIRE(!a->gn_quads || splitk > 1, AZ_E_UNSUPPORTED);
  p.a.gn_quads = nullptr;
  p.cblocks = (a->cout_s + W4C - 1) / W4C;
  p.tblocks = (p.ntiles + W4T - 1) / W4T;
  AZ_REQUIRE((int64_t)p.nk * p.cblocks * W4U_STAGE * 4 <= (1ll << 31), AZ_E_SHAPE);
  const int64_t nwg = (int64_t)p.cblocks * p.tblocks;
  hipStream_t st = az_s(stream);
  {
    static std::atomic<uint64_t> lds4{0};
    hipError_t e = az_max_dynamic_lds((const void*)conv_winograd4_kernel, W4_LDS * 4, lds4);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(conv_winograd4_kernel, dim3((unsigned)nwg, (unsigned)splitk), dim3(512), W4_LDS * 4, st, p);
  int rc = az_launch_status();
  if (rc != AZ_OK) return rc;
  if (splitk > 1) {
    ConvP cp;
    cp.a = p.a;
    cp.a.gn_quads = a->gn_quads;
    cp.npix = p.npix;
    rc = launch_splitk_reduce(cp, st);
  }
  return rc;
}

int az_conv2d_winograd4_suggest_splitk(int64_t batch, int32_t hout, int32_t wout, int32_t cout_s, int32_t cin_s) {
  const int64_t tiles = batch * ((hout + 3) / 4) * ((wout + 3) / 4);
  const int64_t blocks = ((tiles + W4T - 1) / W4T) * ((cout_s + W4C - 1) / W4C);
  const int64_t nk = cin_s / W4K;
  int64_t want = (256 + blocks - 1) / blocks;
  int64_t maxs = nk / 32;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > 16) want = 16;
  if (blocks >= 192) want = 1;
  return (int)(want < 1 ? 1 : want);
}

}  // extern "C"
