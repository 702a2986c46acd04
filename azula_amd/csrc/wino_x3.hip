// Winograd F(2x2, 3x3) with the 16 frequency GEMMs on the bf16 matrix pipe at fp32 accuracy (az_conv2d_winograd_x3_f32).
//
//   out tile (2x2) = A^T [ sum_ci U[xi,nu][co][ci] * V[xi,nu][ci] ] A ,  U = G g G^T (offline),  V = B^T d B
//
// Same algorithm, same transforms and the same fused epilogue as conv_winograd_kernel (conv.hip); what changes is the
// arithmetic of the frequency GEMMs: U (at pack time) and V (in the gather role, after B^T d B) are split EXACTLY into three
// bf16 pieces (common.h: az_split3) and a product is the six largest of the nine partial products on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 6 matrix instructions of 32 cycles per 16 channels where the fp32 form
// issues 8 of 64 cycles (0.375 x the matrix-pipe time; more accurate against fp64 than v_mfma_f32_32x32x2_f32, see
// tests/test_gpu_kernels.py::test_conv2d_x3_accuracy for the direct kernels).
//
// Block = 64 couts x 64 tiles (256 output pixels) x 16 frequencies, K step = 16 input channels, 8 waves = 2 per SIMD.
//   * Wave w = (frequency row xi = w & 3, k = w >> 2) owns the two frequencies (xi, k) and (xi, k + 2) of ALL 64 couts x 64 tiles:
//     8 accumulators acc[4 f + 2 (cout half) + (tile half)].  Every filter fragment and every V fragment of the block is then
//     read by exactly one wave (round 5's first form, wave = 32 couts x 64 tiles x 4 frequencies, read every V fragment twice).
//   * U never passes through LDS: az_winograd_pack_filter_x3_f32 stores it pre-split in MFMA A-fragment order
//     [step][cout block][wave][f][cout half][piece][lane][8 bf16], twelve contiguous 1 KB loads per wave and step.
//   * V = B^T d B lives in LDS as fp32, [frequency][tile][16 channels] rows of 64 B (the 16-byte chunks of a row permuted by
//     (tile >> 2) & 3: conflict-free ds_read_b128 / ds_write_b64), and is split into its three bf16 pieces by the ONE wave
//     that consumes it, in registers, on the way into the MFMAs: 4 B instead of 6 B per value through LDS, no second read.
//     The pipeline runs on HALF-stages of 8 frequencies (nu in {0, 1} | nu in {2, 3}; 32 KB each, two buffers):
//         phase 0: MFMAs on (step s, nu = k) from buffer 0     | V(s, nu 2..3) -> buffer 1, raw pixels of step s + 1 staged
//         phase 1: MFMAs on (step s, nu = k + 2) from buffer 1 | B^T d of step s + 1, V(s + 1, nu 0..1) -> buffer 0
//     one barrier per phase.
//   * The raw input is STAGED in LDS once per step, each pixel once: the 64 tiles of a block are runs ("segments") of
//     horizontally adjacent tiles whose 4x4 patches overlap by two columns, so a segment of n tiles needs 4 rows x (2 n + 2)
//     pixels instead of 16 n.  The pixels' 16 channels of the step (64 B) go global -> registers -> LDS as 16-byte lane slots
//     (bounds-checked buffer loads: padding / ragged tiles land as zeros; upsampling, two sources, circular padding are
//     offsets); then thread (tile tid >> 3, channel pair tid & 7) reads its 4x4 patch with 16 ds_read_b64.
//   * Epilogue: every wave applies the nu side of A^T M A to its two frequencies in registers, the k = 1 waves hand their partial
//     to the k = 0 waves through LDS, which park Z[xi][px] as [xi][tile][px][cout]; all 8 waves read rows back applying the xi
//     side; bias / SiLU / gate / residual / split-K slabs / GroupNorm moments of the output as in the fp32 kernel.
//   The kernel sits at the 1400 W cap (profiles/r05_wino_x3_gate.txt): its time is its energy, so what counts is bytes moved
//   and instructions issued, not cycles.
#include "conv_shared.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int XT = 64;                       // tiles per workgroup
constexpr int XC = 64;                       // output channels per workgroup
constexpr int XK = 16;                       // input channels per step
constexpr int X_ROW = XK * 4;                // bytes of one (frequency, tile) row of V: 16 channels fp32
constexpr int X_FREQ = XT * X_ROW;           // 4 KB
constexpr int X_HALF = 8 * X_FREQ;           // one half-stage (8 frequencies): 32 KB
constexpr int X_STAGE = 2 * X_HALF;          // raw staging behind the two half-stage buffers: <= 768 pixel slots x 64 B = 48 KB
constexpr int X_SLOTS = 768;                 // (8 (64 + segments) slots: up to 32 segments, i.e. maps at least 3 pixels wide; the host checks)
// Stride of a staged pixel slot (the step's 16 channels = 64 B).  Measured and not kept: 80 B, meant to spread the patch reads of a
// wave (8 adjacent tiles = slots two apart) over all banks -- a tie to 1 % slower on five layers in both piece forms, and
// SQ_LDS_BANK_CONFLICT ROSE (35.7 M -> 57.9 M cycles per launch at 4 x 256^2: the 16-byte staging stores then straddle bank groups);
// profiles/r06_wx3_slot_ab.txt, tools/ablate.py wx3_slot80.
constexpr int X_SLOT = 64;
constexpr int X_DOFF1 = X_STAGE + X_SLOTS * X_SLOT;  // the second source's staging offsets (6 per thread, 32 B apart: 16 KB), read back at the switch
constexpr int X_OT = 2 * XC + 4;             // epilogue: floats per tile row of one xi's [tile][px][cout] exchange buffer
constexpr int X_EPI_BYTES = (4 * XT * X_OT + 3 * XT + 384) * 4;  // 137,472 B: four xi partials + tile table + GroupNorm partials
constexpr int X_LDS_BYTES = X_DOFF1 + 512 * 32 > X_EPI_BYTES ? X_DOFF1 + 512 * 32 : X_EPI_BYTES;  // (the epilogue's buffers: >= the K loop's 64 + 48 + 16 KB and the 128 KB hand-off of the k = 1 waves' partials)
static_assert(X_LDS_BYTES >= X_DOFF1 + 512 * 32 && X_LDS_BYTES >= 4 * 128 * 64 * 4, "the epilogue reuses the K loop's LDS");
constexpr int XU_STEP_BYTES = 16 * XC * XK * 3 * 2;  // one (step, cout block) filter chunk: 96 KB
static_assert(4 * X_SLOTS <= 6 * 512, "six staging pieces per thread");
static_assert(X_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// AFF: 0 = plain input, 1 = the input is x * scale + shift (in_affine: a GroupNorm apply pass folded into the gather), 2 = SiLU of that.
// TAIL: a source's channel count is not a multiple of 16 -- the last step of that source masks the channel pairs past its end.
// H2: the "f16x2" form (az_conv2d_winograd_f16x2_f32, include/azula_amd.h): the filter arrives as the IEEE half pieces
//   [uh | ul | uh / 2^11] of U * w_scale in the SAME fragment layout, V is split into two half pieces of V * AZ_F16X2_IN_SCALE,
//   h and l = (V' - h) 2^11 (common.h: az_split2h), and a frequency's product is THREE v_mfma_f32_32x32x16_f16 per 16 channels
//   (ul h + uh h + (uh / 2^11) l) instead of six bf16 ones: 12 matrix instructions per phase, 56 instead of 88 vector
//   instructions of splitting per phase and wave, 8 fragment registers fewer.  The accumulators are multiplied by
//   p.out_scale = 1 / (AZ_F16X2_IN_SCALE * w_scale) (a power of two) behind the K loop; everything else is the same code.
template <int AFF, bool TAIL, bool H2 = false>
__global__ __launch_bounds__(512, 2) void conv_winograd_x3_kernel(WinoP p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  char* const smem = reinterpret_cast<char*>(wsm);
  const AzConvArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;  // 8 waves = 2 per SIMD: wave w and w + 4 share a SIMD
  const int wxi = wave & 3;   // the wave's frequency row xi
  const int wk = wave >> 2;   // its two frequencies: (xi, nu = k) in phase 0, (xi, nu = k + 2) in phase 1
  const int l31 = lane & 31;
  const int h = lane >> 5;

  // workgroup order: see conv_winograd_kernel (contiguous ranges per XCD, rectangles of gt tile blocks x gc cout blocks)
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int rsz = p.gt * p.gc;
  const int rect = wg / rsz, rin = wg - rect * rsz;
  const int rcols = p.cblocks / p.gc;
  const int rrow = rect / rcols;
  const int rin_c = rin / p.gt;
  const int tb = rrow * p.gt + (rin - rin_c * p.gt);
  const int cb = (rect - rrow * rcols) * p.gc + rin_c;
  const int t0 = tb * XT;
  const int tiles_img = p.tiles_h * p.tiles_w;
  const int b_first = t0 / tiles_img;

  const int kt_begin = blockIdx.y * p.kps;
  const int kt_end = min(p.nk, kt_begin + p.kps);
  // H2: the activation scale of this launch: the fixed one, or from the sources' absmax slots (x 4: V = B^T d B sums four pixels)
  const float pin = H2 ? az_f16x2_in_scale(a.in_absmax0, a.in_absmax1, 4.f, lane) : 1.f;
  const float pin2k = pin * 2048.f;

  const int64_t s0_elems = (int64_t)a.h0 * a.w0 * a.c0s;
  const int64_t s1_elems = (int64_t)a.h1 * a.w1 * a.c1s;
  const int b_base = az_depth_base(a, b_first);
  auto clamp_bytes = [](int64_t e) { return (unsigned)(e <= 0 ? 0 : (e * 4 > AZ_RSRC_CLAMP ? AZ_RSRC_CLAMP : e * 4)); };
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.weight, 0, (unsigned)((int64_t)p.nk * p.cblocks * XU_STEP_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src0 + b_base * s0_elems), 0, clamp_bytes((a.batch - b_base) * s0_elems), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 + b_base * s1_elems : a.src0), 0, a.src1 ? clamp_bytes((a.batch - b_base) * s1_elems) : 0u, 0x00020000);

  // ---- gather role (every thread): tile vj, channel pair vq of the step's 16
  const int vj = tid >> 3;
  const int vq = tid & 7;
  int v_b = -1, v_ih0 = 0, v_iw0 = 0;
  bool v_dok = true;
  {
    const int t = t0 + vj;
    if (t < p.ntiles) {
      const int b = t / tiles_img;
      const int r = t - b * tiles_img;
      const int th = r / p.tiles_w;
      v_b = b - b_first;
      v_ih0 = 2 * th - 1;
      v_iw0 = 2 * (r - th * p.tiles_w) - 1;
      (void)az_depth_plane(a, b, v_dok);  // (only whether the depth tap hits a plane: the staging computes the plane itself)
    }
  }
  // ---- staging geometry.  Tile j of the block sits in segment g = (j + tw0) / tiles_w (tw0 = tile column of tile 0; tile rows
  // follow each other seamlessly across images); segment g = tiles [js, js + len) of one tile row, staged as 4 rows of
  // RS = 2 len + 2 pixels from slot S = 8 (js + g) on: slot(j, r, c) = S + r RS + 2 (j - js) + c.  Slots of the block: 8 (64 + G).
  const int tw0 = (t0 % tiles_img) % p.tiles_w;
  const int nseg = (XT - 1 + tw0) / p.tiles_w + 1;
  const int nslots = 8 * (XT + nseg);
  const int ndma = (4 * nslots + 511) >> 9;  // staging pieces per thread and step (one 16-byte lane slot each): 5 or 6 (nslots <= 768)
  auto seg_of = [&](int g, int& js, int& len) {
    js = max(0, g * p.tiles_w - tw0);
    len = min(XT, (g + 1) * p.tiles_w - tw0) - js;
  };
  auto fdiv = [](int x, int d) {  // floor(x / d) for 0 <= x < 2^23, d > 0, without the integer-division expansion
    int q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
    int r = x - q * d;
    q += r >= d ? 1 : (r < 0 ? -1 : 0);
    return q;
  };
  // byte offsets of this thread's lane slots L = m * 512 + tid (pixel slot L >> 2, channels 4 (L & 3) .. + 3 of the step) in source `src`
  unsigned doff[6];
  auto dma_offsets = [&](int src) {
    const int cs = src ? a.c1s : a.c0s;
    const int up = src ? a.up1 : a.up0;
    const int hs = src ? a.h1 : a.h0;
    const int ws = src ? a.w1 : a.w0;
    const int row0 = t0 / p.tiles_w;  // global tile row (image * tiles_h + tile row) of tile 0
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int L = m * 512 + tid;
      const int s = L >> 2;
      const int g = fdiv((s >> 3) + tw0, p.tiles_w + 1);
      int js, len;
      seg_of(g, js, len);
      const int RS = 2 * len + 2;
      const int o = s - 8 * (js + g);
      const int r = fdiv(o, RS);
      const int x = o - r * RS;
      const int grow = row0 + g;
      const int b = fdiv(grow, p.tiles_h);
      const int th = grow - b * p.tiles_h;
      const int tws = g == 0 ? tw0 : 0;
      bool dok = true;
      const int bs = az_depth_plane(a, b, dok) - b_base;
      const int ih = wrap_coord(2 * th - 1 + r, a.hin, a.pad_mode);
      const int iw = wrap_coord(2 * tws - 1 + x, a.win, a.pad_mode);
      const bool ok = s < nslots && b < a.batch && dok && (unsigned)ih < (unsigned)a.hin && (unsigned)iw < (unsigned)a.win;
      doff[m] = ok ? (unsigned)((((bs * hs + (ih >> up)) * ws + (iw >> up)) * cs + (L & 3) * 4) * 4) : OOB;
    }
  };
  uint4* const park1 = reinterpret_cast<uint4*>(smem + X_DOFF1 + tid * 32);
  int cur_src = 0;
  auto switch_source = [&]() __attribute__((always_inline)) {  // (a thread reads back only what it parked itself: no barrier)
    cur_src = 1;
    const uint4 lo = park1[0], hi = park1[1];
    doff[0] = lo.x, doff[1] = lo.y, doff[2] = lo.z, doff[3] = lo.w, doff[4] = hi.x, doff[5] = hi.y;
  };
  if (a.src1) {
    dma_offsets(1);
    park1[0] = make_uint4(doff[0], doff[1], doff[2], doff[3]);
    park1[1] = make_uint4(doff[4], doff[5], 0u, 0u);
  }
  dma_offsets(0);
  if (kt_begin >= p.nkc0) switch_source();  // (a split-K slice inside the second source)
  // the thread's own patch: LDS address of (row 0, column 0), row stride; validity of its 16 positions (in_affine keeps padding at zero)
  int pj_s, pj_len;
  seg_of(fdiv(vj + tw0, p.tiles_w), pj_s, pj_len);
  const int prow = (2 * pj_len + 2) * X_SLOT;
  const char* const patch = smem + X_STAGE + (8 * (pj_s + fdiv(vj + tw0, p.tiles_w)) + 2 * (vj - pj_s)) * X_SLOT + vq * 8;
  unsigned vmask = 0;
  if constexpr (AFF != 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ih = wrap_coord(v_ih0 + r, a.hin, a.pad_mode), iw = wrap_coord(v_iw0 + c, a.win, a.pad_mode);
        if (v_b >= 0 && v_dok && (unsigned)ih < (unsigned)a.hin && (unsigned)iw < (unsigned)a.win) vmask |= 1u << (r * 4 + c);
      }
  }

  f32x2 rv[16];  // raw 4x4 patch of the channel pair (index = patch row * 4 + column), then B^T d in place
  // piece m of the staging of step kt: this thread's lane slot L = m * 512 + tid (16 bytes) global -> registers (gl), and on to
  // the staging area at byte 16 L (gs) half a phase later.  (LDS-DMA moved the same bytes without registers, but one
  // buffer_load ... lds costs the issuing wave 100 - 180 cycles beside MFMAs: profiles/r05_wx3_timeline_v3.txt.)
  float4 gq[6];
  auto gl = [&](int kt, int m) __attribute__((always_inline)) {
    if (m < 5 || m < ndma) {  // (uniform; a block always has at least 520 slots = 4.06 pieces per thread)
      const bool src1 = kt >= p.nkc0;
      const int kc = src1 ? kt - p.nkc0 : kt;
      const __amdgpu_buffer_rsrc_t r = src1 ? rs1 : rs0;
      unsigned off = doff[m];
      if constexpr (TAIL) off = kc * XK + (tid & 3) * 4 < (src1 ? a.c1s : a.c0s) ? off : OOB;  // (channel quads past the source's end)
      gq[m] = buf_ld4(r, off, (unsigned)(kc * XK * 4));
    }
  };
  auto gs = [&](int m) __attribute__((always_inline)) {
    if (m < 5 || m < ndma) *reinterpret_cast<float4*>(smem + X_STAGE + ((m * 512 + tid) >> 2) * X_SLOT + (tid & 3) * 16) = gq[m];
  };
  auto patch_rows = [&](int kt, int r0, int r1) __attribute__((always_inline)) {  // rows [r0, r1) of the staged patch -> rv
#pragma unroll
    for (int r = r0; r < r1; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) rv[4 * r + c] = *reinterpret_cast<const f32x2*>(patch + r * prow + c * X_SLOT);
  };
  // with in_affine the input is act(x * scale + shift) (padding positions stay zero): applied to the raw patch in place
  // the scale / shift pair of the thread's channel pair for step kt: two bounds-checked 8-byte loads issued well ahead of their
  // use (as a pointer dereference in front of the multiply they put a global round trip on every step: - 2 % on the affine layers).
  // (Measured and not kept: the plain affine in the transformed domain, B^T (s d + t m) B = s B^T d B + t (B^T r)(B^T c)^T for the
  //  validity pattern m = r c^T -- 36 instead of 80 vector instructions per step, the same time.)
  const __amdgpu_buffer_rsrc_t raf = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(AFF != 0 ? a.in_affine : nullptr), 0, AFF != 0 ? (unsigned)(2 * a.batch * a.c0s * 4) : 0u, 0x00020000);
  const unsigned af_voff = (unsigned)(((b_first + (v_b < 0 ? 0 : v_b)) * a.c0s + vq * 2) * 4);
  const unsigned af_shift = (unsigned)(a.batch * a.c0s * 4);  // the shift table sits behind the scale table
  f32x2 af_sc = {0.f, 0.f}, af_sh = {0.f, 0.f};
  auto affine_load = [&](int kt) __attribute__((always_inline)) {
    if constexpr (AFF != 0) {
      const bool kv = !TAIL || kt * XK + vq * 2 < a.c0s;  // (masked pairs hold zeros and stay zero)
      af_sc = buf_ld2(raf, kv ? af_voff : OOB, (unsigned)(kt * XK * 4));
      af_sh = buf_ld2(raf, kv ? af_voff : OOB, (unsigned)(kt * XK * 4) + af_shift);
    }
  };
  auto affine = [&]() __attribute__((always_inline)) {
    if constexpr (AFF != 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        f32x2 v = rv[i] * af_sc + af_sh;
        if constexpr (AFF == 2) v = f32x2{az_silu(v.x), az_silu(v.y)};
        rv[i] = (vmask >> i) & 1u ? v : f32x2{0.f, 0.f};
      }
    }
  };
  // B^T d (rows of the patch) of patch column c, in place
  auto row_transform = [&](int c) __attribute__((always_inline)) {
    const f32x2 d0 = rv[c], d1 = rv[4 + c], d2 = rv[8 + c], d3 = rv[12 + c];
    rv[c] = d0 - d2;
    rv[4 + c] = d1 + d2;
    rv[8 + c] = d2 - d1;
    rv[12 + c] = d1 - d3;
  };
  // columns side of B^T d B of frequency row xi for one half of the frequencies (hs = 0: nu in {0, 1}; 1: nu in {2, 3}), stored
  // as fp32 pairs: row (2 xi + j) of buffer hs, tile vj, channels 2 vq, 2 vq + 1
  const int vrow = vj * X_ROW + (((vq >> 1) ^ ((vj >> 2) & 3)) * 16) + (vq & 1) * 8;
  auto nu_store = [&](int hs, int xi) __attribute__((always_inline)) {
    const f32x2 u0 = rv[4 * xi], u1 = rv[4 * xi + 1], u2 = rv[4 * xi + 2], u3 = rv[4 * xi + 3];
    char* dst = smem + hs * X_HALF + vrow + (2 * xi) * X_FREQ;
    *reinterpret_cast<f32x2*>(dst) = hs == 0 ? u0 - u2 : u2 - u1;
    *reinterpret_cast<f32x2*>(dst + X_FREQ) = hs == 0 ? u1 + u2 : u1 - u3;
  };
  // the wave's filter fragments of (step kt, its frequency f, piece pl): both cout halves, rows = couts ch * 32 + l31,
  // k = 8 h .. 8 h + 7 of the step -- two contiguous 1 KB loads
  const unsigned u_lane = (unsigned)(lane * 16);
  // H2: the third piece of a filter value, uh / 2^11, is the first with another exponent: derived here by four v_pk_mul_f16 per
  // fragment (the same rounding as the packing's, bit for bit) instead of loaded -- 8 instead of 12 KB of filter per wave and step
  // through the vector memory path, which a step's 144 KB (filter + gather) load as much as its matrix instructions load the pipe
  constexpr bool H2_DERIVE = true;
  uint4 ua[2][3];  // [cout half][piece]
  typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
  auto derive_u2 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) ua[ch][2] = __builtin_bit_cast(uint4, __builtin_bit_cast(f16x8v, ua[ch][0]) * (_Float16)0.00048828125f);
  };
  auto load_u = [&](int kt, int f, int pl) __attribute__((always_inline)) {
    const unsigned soff =  // (wave-uniform, but derived from threadIdx: readfirstlane makes it a scalar operand)
        (unsigned)__builtin_amdgcn_readfirstlane((int)((((int64_t)kt * p.cblocks + cb) * 8 + wave) * (12 * 1024) + f * (6 * 1024)));
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) ua[ch][pl] = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff));
  };

  f32x16 acc[8];  // [4 f + 2 ch + th]
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  // V fragment of the wave's frequency in a half-stage (row 2 xi + k): tile th * 32 + l31, channels 8 h .. 8 h + 7 = chunks 2 h,
  // 2 h + 1 of the row (chunk c sits at c ^ ((tile >> 2) & 3)), fp32 -> split here into the three bf16 fragments fw[th][piece]
  const int fragB = (2 * wxi + wk) * X_FREQ + l31 * X_ROW;
  const int fsw = (l31 >> 2) & 3;
  float xf[2][8];      // the raw values, then their remainders
  unsigned fw[2][3][4];
  auto frag_read = [&](int hs, int th) __attribute__((always_inline)) {
    const char* vb = smem + hs * X_HALF + fragB + th * (32 * X_ROW);
    const float4 lo = *reinterpret_cast<const float4*>(vb + (((2 * h) ^ fsw) * 16));
    const float4 hi = *reinterpret_cast<const float4*>(vb + (((2 * h + 1) ^ fsw) * 16));
    xf[th][0] = lo.x, xf[th][1] = lo.y, xf[th][2] = lo.z, xf[th][3] = lo.w;
    xf[th][4] = hi.x, xf[th][5] = hi.y, xf[th][6] = hi.z, xf[th][7] = hi.w;
  };
  auto piece = [&](int th, int pl) __attribute__((always_inline)) {  // piece pl = the high halves of the current remainders
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fw[th][pl][j] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, xf[th][2 * j + 1]), __builtin_bit_cast(unsigned, xf[th][2 * j]), 0x07060302u);
  };
  auto remainder = [&](int th, int j0, int j1) __attribute__((always_inline)) {  // x -= its high 16 bits (exact), values 2 j0 .. 2 j1 - 1
#pragma unroll
    for (int i = 2 * j0; i < 2 * j1; ++i) {
      const float x = xf[th][i];
      xf[th][i] = x - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u);
    }
  };
  // the six partial products of a frequency in the order the B pieces become available: b0 (3 products), b1 (2), b2 (1); the A
  // pieces free up in the order a2 (after 4 MFMAs), a1 (after 16), a0 -- their registers take the next phase's fragments
  // (H2: b = h serves the products with ul and uh, b = l the one with uh / 2^11; the A pieces free up in the order 1, 0, 2)
  constexpr int PA[6] = {H2 ? 1 : 2, H2 ? 0 : 1, H2 ? 2 : 0, 1, 0, 0};
  constexpr int PB[6] = {0, 0, H2 ? 1 : 0, 1, 1, 2};
  auto mf = [&](int f, int k) __attribute__((always_inline)) {  // MFMA k = 0..23 (H2: 0..11) of a phase: product t = k / 4, tile half, cout half
    const int t = k >> 2, th = (k >> 1) & 1, ch = k & 1;
    f32x16& c = acc[4 * f + 2 * ch + th];
    uint4 bw = make_uint4(fw[th][PB[t]][0], fw[th][PB[t]][1], fw[th][PB[t]][2], fw[th][PB[t]][3]);
    if constexpr (H2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ua[ch][PA[t]]), __builtin_bit_cast(f16x8, bw), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[ch][PA[t]]), __builtin_bit_cast(bf16x8, bw), c, 0, 0, 0);
  };
  // H2: the two half pieces of tile half th (common.h: az_split2h, in two steps so that the products with h can start before the
  // low piece exists): h = fp16(V'), xf <- V' 2^11;  then l = fp16(xf - 2^11 h)
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  auto h_piece = [&](int th) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 x = {xf[th][2 * j], xf[th][2 * j + 1]};
      const f32x2 v = x * pin;
      fw[th][0][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2v));
      const f32x2 t = x * pin2k;
      xf[th][2 * j] = t.x, xf[th][2 * j + 1] = t.y;
    }
  };
  auto l_piece = [&](int th) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const h2v hh = __builtin_bit_cast(h2v, fw[th][0][j]);
      const h2v l = {(_Float16)__builtin_fmaf((float)hh.x, -2048.f, xf[th][2 * j]), (_Float16)__builtin_fmaf((float)hh.y, -2048.f, xf[th][2 * j + 1])};
      fw[th][1][j] = __builtin_bit_cast(unsigned, l);
    }
  };

  // One phase = the 24 MFMAs of (step kt, the wave's frequency f = hs) + the production of the NEXT half-stage into the other buffer:
  //   hs = 0: V(kt, nu 2..3) -> buffer 1 from the B^T d of step kt held in rv; the raw pixels of step ktn are staged;
  //   hs = 1: the thread's patch of step ktn, its B^T d, V(ktn, nu 0..1) -> buffer 0.
  // Issued in SLOTS of one MFMA plus a few other instructions, fenced by sched_barrier(0) (the compiler's own order puts all 24
  // MFMAs first and everything else behind them; a sched_group_barrier pattern was not honoured).  The consumer side of a
  // slot splits the phase's own V fragments just ahead of the products that need them; the producer side follows.
#define XS_FENCE __builtin_amdgcn_sched_barrier(0)
  auto phase = [&](auto HS, int kt, int ktn) __attribute__((always_inline)) {
    constexpr int hs = decltype(HS)::value;
    constexpr int ob = 1 - hs;  // the buffer (and half of the frequencies) this phase produces
    const int ktu = hs == 0 ? kt : ktn;  // the step whose filter fragments are loaded next (frequency f = ob of it)
    // (entry: xf[0], xf[1] hold the phase's raw fragments, read behind the barrier)
    if constexpr (H2) {
      // 12 products per phase; the producer side is the same work as below, two slots' worth per slot
      h_piece(0); XS_FENCE;
      if constexpr (hs == 0) {
        // (the staging loads of the next step as early as rv allows and their LDS stores spread over the last four slots: the
        //  longest cover for the loads -- measured against three other orders, profiles/r06_f16x2_sched_ab.txt: - 2.4 ... - 3 %)
        mf(hs, 0); h_piece(1); XS_FENCE;
        mf(hs, 1); nu_store(ob, 0); nu_store(ob, 1); XS_FENCE;
        mf(hs, 2); nu_store(ob, 2); nu_store(ob, 3); gl(ktn, 0); gl(ktn, 1); gl(ktn, 2); XS_FENCE;  // (rv is free from here)
        mf(hs, 3); gl(ktn, 3); gl(ktn, 4); gl(ktn, 5); load_u(ktu, ob, 1); XS_FENCE;  // (a1 is free: the next phase's a1)
        mf(hs, 4); l_piece(0); XS_FENCE;
        mf(hs, 5); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;
        mf(hs, 6); XS_FENCE;
        mf(hs, 7); load_u(ktu, ob, 0); XS_FENCE;  // (a0 is free)
        mf(hs, 8); gs(0); XS_FENCE;
        mf(hs, 9); gs(1); gs(2); XS_FENCE;
        mf(hs, 10); gs(3); gs(4); XS_FENCE;
        mf(hs, 11); gs(5);
      } else {
        mf(hs, 0); h_piece(1); affine_load(ktn); XS_FENCE;
        mf(hs, 1); patch_rows(ktn, 0, 2); XS_FENCE;
        mf(hs, 2); patch_rows(ktn, 2, 4); XS_FENCE;
        mf(hs, 3); l_piece(0); load_u(ktu, ob, 1); XS_FENCE;
        mf(hs, 4); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;
        affine();
        mf(hs, 5); row_transform(0); row_transform(1); XS_FENCE;
        mf(hs, 6); row_transform(2); row_transform(3); XS_FENCE;
        mf(hs, 7); load_u(ktu, ob, 0); XS_FENCE;
        mf(hs, 8); nu_store(ob, 0); XS_FENCE;
        mf(hs, 9); nu_store(ob, 1); XS_FENCE;
        mf(hs, 10); nu_store(ob, 2); XS_FENCE;
        mf(hs, 11); nu_store(ob, 3);
      }
      if constexpr (!H2_DERIVE) load_u(ktu, ob, 2);
    } else {
    piece(0, 0); XS_FENCE;
    if constexpr (hs == 0) {
      // (rv holds B^T d of step kt: consumed first, its registers then take the staging loads; the staging area is free: every
      //  thread read its patch in the previous phase 1)
      mf(hs, 0); piece(1, 0); XS_FENCE;
      mf(hs, 1); nu_store(ob, 0); XS_FENCE;
      mf(hs, 2); nu_store(ob, 1); XS_FENCE;
      mf(hs, 3); nu_store(ob, 2); XS_FENCE;
      mf(hs, 4); nu_store(ob, 3); load_u(ktu, ob, 2); XS_FENCE;  // (rv is free from here; a2 is free: the next phase's a2)
      mf(hs, 5); remainder(0, 0, 2); XS_FENCE;
      mf(hs, 6); remainder(0, 2, 4); XS_FENCE;
      mf(hs, 7); remainder(1, 0, 2); XS_FENCE;
      mf(hs, 8); remainder(1, 2, 4); XS_FENCE;
      mf(hs, 9); piece(0, 1); XS_FENCE;
      mf(hs, 10); piece(1, 1); XS_FENCE;
      mf(hs, 11); gl(ktn, 0); gl(ktn, 1); remainder(0, 0, 2); XS_FENCE;
      mf(hs, 12); gl(ktn, 2); gl(ktn, 3); remainder(0, 2, 4); XS_FENCE;
      mf(hs, 13); gl(ktn, 4); gl(ktn, 5); remainder(1, 0, 2); XS_FENCE;
      mf(hs, 14); remainder(1, 2, 4); XS_FENCE;
      mf(hs, 15); piece(0, 2); piece(1, 2); XS_FENCE;
      mf(hs, 16); load_u(ktu, ob, 1); XS_FENCE;  // (a1 is free)
      mf(hs, 17); gs(0); XS_FENCE;
      mf(hs, 18); gs(1); XS_FENCE;
      mf(hs, 19); gs(2); XS_FENCE;
      mf(hs, 20); gs(3); XS_FENCE;
      mf(hs, 21); gs(4); XS_FENCE;
      mf(hs, 22); gs(5); XS_FENCE;
      mf(hs, 23);
    } else {
      // (the raw pixels of step ktn are staged: behind the barrier that closed phase 0; the patch is read late, when the
      //  fragment registers of this phase are mostly done)
      mf(hs, 0); piece(1, 0); XS_FENCE;
      mf(hs, 1); affine_load(ktn); remainder(0, 0, 2); XS_FENCE;
      mf(hs, 2); remainder(0, 2, 4); XS_FENCE;
      mf(hs, 3); remainder(1, 0, 2); XS_FENCE;
      mf(hs, 4); remainder(1, 2, 4); load_u(ktu, ob, 2); XS_FENCE;  // (a2 is free: the next phase's a2)
      mf(hs, 5); piece(0, 1); XS_FENCE;
      mf(hs, 6); piece(1, 1); XS_FENCE;
      mf(hs, 7); remainder(0, 0, 2); XS_FENCE;
      mf(hs, 8); remainder(0, 2, 4); XS_FENCE;
      mf(hs, 9); remainder(1, 0, 2); XS_FENCE;
      mf(hs, 10); remainder(1, 2, 4); XS_FENCE;
      mf(hs, 11); piece(0, 2); piece(1, 2); XS_FENCE;
      mf(hs, 12); patch_rows(ktn, 0, 2); XS_FENCE;
      mf(hs, 13); patch_rows(ktn, 2, 4); XS_FENCE;
      mf(hs, 14); XS_FENCE;
      mf(hs, 15); XS_FENCE;
      mf(hs, 16); load_u(ktu, ob, 1); XS_FENCE;  // (a1 is free)
      affine();
      mf(hs, 17); row_transform(0); row_transform(1); XS_FENCE;
      mf(hs, 18); row_transform(2); row_transform(3); XS_FENCE;
      mf(hs, 19); nu_store(ob, 0); XS_FENCE;
      mf(hs, 20); nu_store(ob, 1); XS_FENCE;
      mf(hs, 21); nu_store(ob, 2); XS_FENCE;
      mf(hs, 22); nu_store(ob, 3); XS_FENCE;
      mf(hs, 23);
    }
    load_u(ktu, ob, 0);
    }
    // the NEXT phase's fragments: its buffer is complete behind the barrier.  (Measured and not kept -- profiles/r05_wx3_sched_ab.txt:
    // the barrier AHEAD of the last four products, to cover the LDS round trip of these reads with matrix work, and the producer
    // side three slots earlier: + 3.5 % cycles, a wave then waits with products unissued.)
    __syncthreads();
    frag_read(ob, 0); frag_read(ob, 1);
    XS_FENCE;
  };

  if (kt_begin < kt_end) {
#pragma unroll
    for (int m = 0; m < 6; ++m) gl(kt_begin, m);
    // (piece 2 first, fenced: the order in which the loop keeps its filter loads in flight.  The wait counts the compiler derives
    //  at the loop header are the stricter of the entry's and the back edge's; with the pieces in ascending order the entry made
    //  the first two products of EVERY step wait for all outstanding filter loads -- vmcnt(2) / vmcnt(0) instead of 5 / 4)
    if constexpr (H2) {
      XS_FENCE; load_u(kt_begin, 0, 1); XS_FENCE; load_u(kt_begin, 0, 0); XS_FENCE;
      if constexpr (!H2_DERIVE) { load_u(kt_begin, 0, 2); XS_FENCE; }
    } else {
      XS_FENCE; load_u(kt_begin, 0, 2); XS_FENCE; load_u(kt_begin, 0, 1); XS_FENCE; load_u(kt_begin, 0, 0); XS_FENCE;
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) gs(m);
    __syncthreads();
    affine_load(kt_begin);
    patch_rows(kt_begin, 0, 4);
    affine();
#pragma unroll
    for (int c = 0; c < 4; ++c) row_transform(c);
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) nu_store(0, xi);
    __syncthreads();
    frag_read(0, 0); frag_read(0, 1);
    // waves 4 .. 7 are the younger half of every SIMD pair and lose the vector-issue arbitration by age: static priority
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int ktn = min(kt + 1, kt_end - 1);  // (the last step restages itself into a buffer nobody reads: no branch in the body)
      if (ktn >= p.nkc0 && cur_src == 0) switch_source();  // (uniform, once per K walk: this iteration gathers from the second source)
      phase(std::integral_constant<int, 0>{}, kt, ktn);
      phase(std::integral_constant<int, 1>{}, kt, ktn);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();
#undef XS_FENCE
  if constexpr (H2) {  // back to the operands' scale (exact: powers of two)
    const float osc = __fdiv_rn(p.out_scale, pin);
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = acc[q] * osc;
  }

  // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  Z[xi][px] = sum_nu M[xi][nu] A[nu][px]: px = 0: M0 + M1 + M2,
  // px = 1: M1 - M2 - M3.  A wave holds M[xi][k] (f = 0) and M[xi][k + 2] (f = 1) of all 64 couts x 64 tiles:
  //   k = 0: P0 = M0 + M2, P1 = -M2;   k = 1: P0 = M1, P1 = M1 - M3;   Z[px] = P_px(k = 0) + P_px(k = 1).
  // The k = 1 waves hand (P0, P1) to the k = 0 wave of their xi through LDS (same lane, same register: 32 float4 per lane, lane-
  // contiguous), which adds and parks Z in LDS as [xi][tile][px][cout] (tile stride padded by 4 floats: conflict-free
  // ds_write_b128).  Then ALL 8 waves read rows back, applying the xi side on the way (py = 0: Z[0] + Z[1] + Z[2]; py = 1:
  // Z[1] - Z[2] - Z[3]), so that 16 consecutive lanes store the 256 contiguous bytes of one output pixel.
  // Lane: tile = th * 32 + l31; couts ch * 32 + 8 g + 4 h + (0..3) in registers 4 g .. 4 g + 3 of acc[4 f + 2 ch + th].
  // (two code versions behind a scalar branch on the wave-uniform k: as selects the compiler computes both and spills)
  if (wk == 0) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q] = acc[q] + acc[4 + q];  // P0 = M0 + M2
      acc[4 + q] = -acc[4 + q];      // P1 = -M2
    }
  } else {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[4 + q] = acc[q] - acc[4 + q];  // P0 = M1 (in place), P1 = M1 - M3
  }
  {
    float4* hand = reinterpret_cast<float4*>(wsm) + wxi * (32 * 64) + lane;  // [xi][32 float4][lane]
    if (wk == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) hand[(q * 4 + g) * 64] = make_float4(acc[q][4 * g], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]);
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 o = hand[(q * 4 + g) * 64];
          acc[q][4 * g] += o.x, acc[q][4 * g + 1] += o.y, acc[q][4 * g + 2] += o.z, acc[q][4 * g + 3] += o.w;
        }
    }
    __syncthreads();  // (the hand-off area is dead: Z goes on top of it)
    if (wk == 0) {
      float* zbuf = wsm + wxi * (XT * X_OT);
#pragma unroll
      for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          float* zrow = zbuf + (th * 32 + l31) * X_OT + ch * 32 + 4 * h;
          const f32x16& z0 = acc[2 * ch + th];
          const f32x16& z1 = acc[4 + 2 * ch + th];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(zrow + 8 * g) = make_float4(z0[4 * g], z0[4 * g + 1], z0[4 * g + 2], z0[4 * g + 3]);
            *reinterpret_cast<float4*>(zrow + XC + 8 * g) = make_float4(z1[4 * g], z1[4 * g + 1], z1[4 * g + 2], z1[4 * g + 3]);
          }
        }
    }
  }
  int* tinfo = reinterpret_cast<int*>(wsm + 4 * XT * X_OT);  // [64] first output pixel of the tile (-1: none), [64] flags, [64] image
  if (wave < 2 && h == 0) {
    const int tl = wave * 32 + l31;
    const int t = t0 + tl;
    int n00 = -1, fl = 0, b = 0;
    if (t < p.ntiles) {
      b = t / tiles_img;
      const int rr = t - b * tiles_img;
      const int th = rr / p.tiles_w;
      const int tw = rr - th * p.tiles_w;
      n00 = (b * a.hout + 2 * th) * a.wout + 2 * tw;
      fl = (2 * th + 1 < a.hout ? 1 : 0) | (2 * tw + 1 < a.wout ? 2 : 0);
    }
    tinfo[tl] = n00;
    tinfo[XT + tl] = fl;
    tinfo[2 * XT + tl] = b;
  }
  __syncthreads();
  const int cq = tid & 15;  // the same channel quad in every iteration
  const int co = cb * XC + cq * 4;
  if (co >= a.cout_s) return;
  int on[8], ob[8];
  float4 ov[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = (it * 512 + tid) >> 4;  // tile * 4 + pixel
    const int tile = row >> 2, px = row & 3;
    const int n00 = tinfo[tile], fl = tinfo[XT + tile];
    ob[it] = tinfo[2 * XT + tile];
    const bool skip = n00 < 0 || (((px >> 1) & ~fl) | ((px & 1) & ~(fl >> 1)));
    on[it] = skip ? -1 : n00 + (px >> 1) * a.wout + (px & 1);
    const float* zp = wsm + tile * X_OT + (px & 1) * XC + cq * 4;
    // py = 0: (z1 + z2) + z0;  py = 1: (z1 - z2) - z3 -- one code path for both: the sign as an FMA factor (exact), the third
    // term's buffer by address
    const int py = px >> 1;
    const float sg = py ? -1.f : 1.f;
    const float4 z1 = *reinterpret_cast<const float4*>(zp + 1 * (XT * X_OT));
    const float4 z2 = *reinterpret_cast<const float4*>(zp + 2 * (XT * X_OT));
    const float4 za = *reinterpret_cast<const float4*>(zp + (py ? 3 : 0) * (XT * X_OT));
    ov[it] = make_float4(fmaf(sg, za.x, fmaf(sg, z2.x, z1.x)), fmaf(sg, za.y, fmaf(sg, z2.y, z1.y)),
                         fmaf(sg, za.z, fmaf(sg, z2.z, z1.z)), fmaf(sg, za.w, fmaf(sg, z2.w, z1.w)));
  }
  if (a.gn_quads == nullptr) {
    epilogue_store_batch<8>(a, on, ob, co, ov, (int64_t)blockIdx.y * p.npix);
    return;
  }
  // ---- GroupNorm statistics of the OUTPUT (see conv_winograd_kernel: one partial per (image, tile block, channel quad); the host
  // enables this only when the 64 tiles of a workgroup lie in one image, whole 2 x 2 tiles, cout_s % 64 == 0)
  float mom[3] = {0.f, 0.f, 0.f};
  epilogue_store_batch<8, true>(a, on, ob, co, ov, (int64_t)blockIdx.y * p.npix, mom);
  float am = mom[0] + mom[1] * (1.f / 32.f);
  float a2 = mom[2] - mom[1] * mom[1] * (1.f / 32.f);
  auto chan = [](float& am, float& a2, float bm, float b2, float half_n) {  // both sides hold 2 * half_n values
    const float d = bm - am;
    am = am + 0.5f * d;
    a2 = (a2 + b2) + d * d * half_n;
  };
  chan(am, a2, __shfl_xor(am, 16), __shfl_xor(a2, 16), 16.f);
  chan(am, a2, __shfl_xor(am, 32), __shfl_xor(a2, 32), 32.f);
  float* sh = reinterpret_cast<float*>(tinfo + 3 * XT);
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
    const int chunk = (t0 - b_first * tiles_img) / XT;
    float* out = a.gn_quads + ((((int64_t)b_first * a.gn_chunks + chunk) * (a.cout_s / 4)) + (cb * (XC / 4) + tid)) * 4;
    out[0] = 1024.f;
    out[1] = wm[0];
    out[2] = w2[0];
    out[3] = 0.f;
  }
}

}  // namespace

// Launch (host side of az_conv2d_winograd_x3_f32, conv.hip validates the descriptor and fills `p` in 16-channel steps).
template <int AFF, bool TAIL, bool H2>
static int launch_x3(const WinoP& p, unsigned splitk, hipStream_t st) {
  static std::atomic<uint64_t> lds_set{0};  // (one per instantiation, one bit per device: common.h)
  hipError_t e = az_max_dynamic_lds((const void*)conv_winograd_x3_kernel<AFF, TAIL, H2>, X_LDS_BYTES, lds_set);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv_winograd_x3_kernel<AFF, TAIL, H2>), dim3((unsigned)((int64_t)p.cblocks * p.tblocks), splitk), dim3(512), X_LDS_BYTES,
                     st, p);
  return az_launch_status();
}

template <bool H2>
static int launch_x3_mode(const WinoP& p, unsigned splitk, hipStream_t st) {
  const bool tail = (p.a.c0s % XK) != 0 || (p.a.c1s % XK) != 0;
  const int aff = !p.a.in_affine ? 0 : (p.a.in_act == 0 ? 1 : 2);
  switch (aff * 2 + (tail ? 1 : 0)) {
    case 0: return launch_x3<0, false, H2>(p, splitk, st);
    case 1: return launch_x3<0, true, H2>(p, splitk, st);
    case 2: return launch_x3<1, false, H2>(p, splitk, st);
    case 3: return launch_x3<1, true, H2>(p, splitk, st);
    case 4: return launch_x3<2, false, H2>(p, splitk, st);
    default: return launch_x3<2, true, H2>(p, splitk, st);
  }
}

// h2: the f16x2 form (p.out_scale set by the caller)
__attribute__((visibility("hidden"))) int azi_winograd_x3_launch(const WinoP& p, unsigned splitk, hipStream_t st, bool h2) {
  return h2 ? launch_x3_mode<true>(p, splitk, st) : launch_x3_mode<false>(p, splitk, st);
}
