r"""Timing ablations WITHOUT ablation code in the product sources: builds a variant of libazula_amd.so from a patched COPY of
azula_amd/csrc (textual substitutions listed below; results of such a library are WRONG, only its timing means something).

    python tools/ablate.py NAME [NAME ...]     ->  azula_amd/csrc/_ab/libazula_amd_NAME.so   (select with AZULA_AMD_LIB=<path>)
    python tools/ablate.py --list
    python tools/ablate.py --head FILE [FILE ...]   ->  .../_ab/libazula_amd_head.so: the named sources as committed at HEAD, the
                                                        rest from the working tree (A/B of an uncommitted kernel change)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from azula_amd.csrc import build as B  # noqa: E402

# name -> [(file, old, new), ...]; every `old` must occur exactly once
VARIANTS = {
    # ---- wino_x3.hip (the second form of its K loop; profiles/r05_wx3_ablation_v2.txt and r05_wx3_energy_probes.txt were measured
    #      with the corresponding patches of the first form: git history of this file)
    # no-load variants FREEZE operands (constant data toggles nothing in the matrix pipe): they overstate under the power cap
    "wx3_nogather": [("wino_x3.hip", "      gq[m] = buf_ld4(r, off, (unsigned)(kc * XK * 4));",
                      "      asm volatile(\"\" : \"+v\"(gq[m].x), \"+v\"(gq[m].y), \"+v\"(gq[m].z), \"+v\"(gq[m].w) : \"v\"(off));")],
    "wx3_nou": [("wino_x3.hip", "    for (int ch = 0; ch < 2; ++ch) ua[ch][pl] = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff));",
                 "    for (int ch = 0; ch < 2; ++ch) asm volatile(\"\" : \"+v\"(ua[ch][pl]) : \"s\"(soff));")],
    "wx3_nomfma": [("wino_x3.hip", "    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[ch][PA[t]]), __builtin_bit_cast(bf16x8, bw), c, 0, 0, 0);",
                    "    else asm volatile(\"\" : \"+v\"(c) : \"v\"(bw.x), \"v\"(bw.y), \"v\"(bw.z), \"v\"(bw.w));")],
    # energy probes on RANDOM data (the kernel sits at the 1400 W cap: time = energy / cap, so a variant's time change is the
    # energy share of what it removes, as long as the data stay random)
    "wx3_halfmfma": [("wino_x3.hip", "    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[ch][PA[t]]), __builtin_bit_cast(bf16x8, bw), c, 0, 0, 0);",
                      "    else if (t < 3) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[ch][PA[t]]), __builtin_bit_cast(bf16x8, bw), c, 0, 0, 0);")],
    "wx3_nosplit": [("wino_x3.hip", "      xf[th][i] = x - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u);",
                     "      xf[th][i] = x;")],
    "wx3_halfstore": [("wino_x3.hip", "    *reinterpret_cast<f32x2*>(dst + X_FREQ) = hs == 0 ? u1 + u2 : u1 - u3;",
                       "    asm volatile(\"\" :: \"v\"(hs == 0 ? u1 + u2 : u1 - u3));")],
}
# attention_x3_kernel: the running maximum updated (and the accumulators rescaled) in every tile, as before the lazy form
VARIANTS["att_eager"] = [("attention.hip", "      const bool jump = mt > m_run + 8.f;", "      const bool jump = mt > m_run;")]
# attention_x3_kernel timing ablations (64 x 12 heads x 256 tokens x 64: profiles/r05_attention_x3_ablation.txt)
_ATT_CHEAP = "{{ {0} = __builtin_bit_cast(unsigned, {3}); {1} = {0}; {2} = {0}; }}"
VARIANTS["att_nopsplit"] = [("attention.hip", "          else az_split3(sacc[8 * s2 + 2 * j], sacc[8 * s2 + 2 * j + 1], p3[0][j], p3[1][j], p3[2][j]);",
                             "          else " + _ATT_CHEAP.format("p3[0][j]", "p3[1][j]", "p3[2][j]", "sacc[8 * s2 + 2 * j]"))]
VARIANTS["att_nokvsplit"] = [
    ("attention.hip", "          az_split3(kv.x, kv.y, k3[0][0], k3[1][0], k3[2][0]);\n          az_split3(kv.z, kv.w, k3[0][1], k3[1][1], k3[2][1]);\n"
                      "          az_split3(vv.x, vv.y, v3[0][0], v3[1][0], v3[2][0]);\n          az_split3(vv.z, vv.w, v3[0][1], v3[1][1], v3[2][1]);\n",
     "        " + _ATT_CHEAP.format("k3[0][0]", "k3[1][0]", "k3[2][0]", "kv.x") + _ATT_CHEAP.format("k3[0][1]", "k3[1][1]", "k3[2][1]", "kv.z")
     + _ATT_CHEAP.format("v3[0][0]", "v3[1][0]", "v3[2][0]", "vv.x") + _ATT_CHEAP.format("v3[0][1]", "v3[1][1]", "v3[2][1]", "vv.z") + "\n")]
VARIANTS["att_1mfma"] = [("attention.hip", "for (int t = 0; t < (H2 ? 3 : 6); ++t) {", "for (int t = 0; t < 1; ++t) {"),
                         ("attention.hip", "for (int u = 0; u < (H2 ? 3 : 6); ++u) {", "for (int u = 0; u < 1; ++u) {")]
VARIANTS["att_noexp"] = [("attention.hip", "const float pe = __builtin_amdgcn_exp2f(sacc[r] - m_sub);", "const float pe = sacc[r] - m_sub;")]
VARIANTS["att_nosoftmax"] = VARIANTS["att_noexp"] + VARIANTS["att_nopsplit"]
VARIANTS["att_novalu"] = VARIANTS["att_nosoftmax"] + VARIANTS["att_nokvsplit"]
VARIANTS["att_mfmaonly"] = VARIANTS["att_novalu"]  # (alias)
VARIANTS["att_all"] = VARIANTS["att_novalu"] + VARIANTS["att_1mfma"]
VARIANTS["wx3_noloads"] = VARIANTS["wx3_nogather"] + VARIANTS["wx3_nou"]

# ---- wino_x3.hip: per-wave phase timeline (s_memtime sums per workgroup, waves 0 and 4 of the first 512 workgroups):
#      sections of a phase: [0] slots 0-11, [1] slots 12-23, [2] last filter loads, [3] barrier + next fragment reads
VARIANTS["wx3_tl"] = [
    ("wino_x3.hip", "namespace {\n\ntypedef __bf16 bf16x8 __attribute__",
     "__device__ unsigned az_wx3_tl[512 * 2 * 12];\n"
     "extern \"C\" int az_debug_wx3_timeline(unsigned* host, int n_words) {\n"
     "  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(az_wx3_tl), (size_t)n_words * 4, 0, hipMemcpyDeviceToHost);\n}\n"
     "namespace {\n\ntypedef __bf16 bf16x8 __attribute__"),
    ("wino_x3.hip", "  uint4 ua[2][3];  // [cout half][piece]\n",
     "  uint4 ua[2][3];\n  unsigned tlacc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};\n  const unsigned long long tl_entry = __builtin_readcyclecounter();\n"),
    ("wino_x3.hip", "    constexpr int ob = 1 - hs;  // the buffer (and half of the frequencies) this phase produces\n",
     "    constexpr int ob = 1 - hs;\n    const unsigned long long tl0 = __builtin_readcyclecounter();\n    unsigned long long tl1 = tl0, tl2 = tl0;\n"),
    ("wino_x3.hip", "      mf(hs, 12); gl(ktn, 2); gl(ktn, 3); remainder(0, 2, 4); XS_FENCE;\n",
     "      tl1 = __builtin_readcyclecounter();\n      mf(hs, 12); gl(ktn, 2); gl(ktn, 3); remainder(0, 2, 4); XS_FENCE;\n"),
    ("wino_x3.hip", "      mf(hs, 12); patch_rows(ktn, 0, 2); XS_FENCE;\n",
     "      tl1 = __builtin_readcyclecounter();\n      mf(hs, 12); patch_rows(ktn, 0, 2); XS_FENCE;\n"),
    ("wino_x3.hip", "    load_u(ktu, ob, 0);\n    }\n    // the NEXT phase's fragments",
     "    tl2 = __builtin_readcyclecounter();\n    load_u(ktu, ob, 0);\n    }\n    const unsigned long long tl3 = __builtin_readcyclecounter();\n    // the NEXT phase's fragments"),
    ("wino_x3.hip", "    __syncthreads();\n    frag_read(ob, 0); frag_read(ob, 1);\n    XS_FENCE;\n  };\n",
     "    __syncthreads();\n    frag_read(ob, 0); frag_read(ob, 1);\n    XS_FENCE;\n"
     "    const unsigned long long tl4 = __builtin_readcyclecounter();\n"
     "    tlacc[hs][0] += (unsigned)(tl1 - tl0); tlacc[hs][1] += (unsigned)(tl2 - tl1); tlacc[hs][2] += (unsigned)(tl3 - tl2); tlacc[hs][3] += (unsigned)(tl4 - tl3);\n  };\n"),
    ("wino_x3.hip", "  __syncthreads();\n#undef XS_FENCE\n",
     "  __syncthreads();\n#undef XS_FENCE\n  const unsigned long long tl_loop = __builtin_readcyclecounter();\n"),
    ("wino_x3.hip", "  if (a.gn_quads == nullptr) {\n    epilogue_store_batch<8>(a, on, ob, co, ov, (int64_t)blockIdx.y * p.npix);\n",
     "  if (a.gn_quads == nullptr) {\n    epilogue_store_batch<8>(a, on, ob, co, ov, (int64_t)blockIdx.y * p.npix);\n"
     "    if ((tid & 255) == 0 && blockIdx.x < 512 && blockIdx.y == 0) {\n"
     "      __builtin_amdgcn_s_waitcnt(0);\n"
     "      unsigned* o = az_wx3_tl + (blockIdx.x * 2 + (tid >> 8)) * 12;\n"
     "      for (int i = 0; i < 8; ++i) o[i] = tlacc[i >> 2][i & 3];\n"
     "      o[8] = (unsigned)(tl_loop - tl_entry); o[9] = (unsigned)(__builtin_readcyclecounter() - tl_loop); o[10] = (unsigned)tl_entry; o[11] = __builtin_amdgcn_s_getreg(63492);\n"
     "    }\n"),
]

# ---- the ablations / A-B switches that lived in the product sources as -D macros until round 4 (conv.hip, attention.hip)
VARIANTS["igemm_noload"] = [("conv.hip", "      load_tile();  // global loads in flight under the MFMAs below\n", "")]
VARIANTS["igemm_nostore"] = [("conv.hip", "    if (more) store_tile(buf ^ 1);\n    __syncthreads();\n  }\n  }\n", "  }\n  }\n")]
VARIANTS["x3_nosplit"] = [  # what would activations that arrive pre-split cost the 128 x 128 bf16x3 kernel?
    ("conv.hip", "        else split3(x[2 * j], x[2 * j + 1], q[0][j], q[1][j], q[2][j]);\n      }\n#pragma unroll\n      for (int pl = 0; pl < (H2 ? 2 : 3); ++pl)\n        *reinterpret_cast<uint4*>(xsm",
     "        else q[0][j] = q[1][j] = q[2][j] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[2 * j + 1]), __builtin_bit_cast(unsigned, x[2 * j]), 0x07060302u);\n      }\n#pragma unroll\n      for (int pl = 0; pl < (H2 ? 2 : 3); ++pl)\n        *reinterpret_cast<uint4*>(xsm"),
]
# ---- conv_gemm_half_big_kernel (half-precision operands, 256 x 256 tile; typed launches): where does a K step's time go?  (timing only;
#      profiles/r06_half_gemm.txt (5) was measured with these on the two-register-set form that was then dropped)
_HB_KS = "#pragma unroll\n    for (int ks = 0; ks < HGK / 16; ++ks) {\n"
_HB_IT = "    store_step(buf ^ 1);                               // step i + 1 (in registers since the previous iteration) -> the other stage\n    load_step(min(kt_begin + i + 2, kt_end - 1));      // in flight under the MFMAs below and the next iteration's first ones\n" + _HB_KS
VARIANTS["hbig_noload"] = [("conv.hip", _HB_IT, "    store_step(buf ^ 1);\n" + _HB_KS)]
VARIANTS["hbig_nostage"] = [("conv.hip", _HB_IT, _HB_KS)]  # neither the stage stores nor the loads: fragment reads + matrix instructions + barrier
VARIANTS["hbig_nomfma"] = [
    ("conv.hip", "          if constexpr (F16) acc[ci][pj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ci], fb[pj], acc[ci][pj], 0, 0, 0);\n"
                 "          else acc[ci][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ci], fb[pj], acc[ci][pj], 0, 0, 0);\n        }\n    }\n    // issue order: the 24 fragment reads in four groups",
     "          asm volatile(\"\" : \"+v\"(acc[ci][pj]) : \"v\"(fa[ci]), \"v\"(fb[pj]));\n        }\n    }\n    // issue order: the 24 fragment reads in four groups")]
VARIANTS["hbig_noepi"] = [("conv.hip", "  if (a.dst_dtype) gemm_big_epilogue<4, F16 ? 2 : 1>(p, acc, m0, n0, wc, wp, lane, tid, gsmf);\n  else gemm_big_epilogue<4>(p, acc, m0, n0, wc, wp, lane, tid, gsmf);\n",
                           "  if (acc[0][0][0] == 12345.678f) gemm_big_epilogue<4>(p, acc, m0, n0, wc, wp, lane, tid, gsmf);\n")]
VARIANTS["hbig_nosched"] = [("conv.hip", "    for (int k = 0; k < 32; ++k) {\n      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);\n      if ((k & 7) == 1 && k < 24) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // the next k-step's fragments",
                             "    for (int k = 0; k < 0; ++k) {\n      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);\n      if ((k & 7) == 1 && k < 24) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // the next k-step's fragments")]
VARIANTS["x3big_nosched"] = [("conv.hip", "    if constexpr (!H2) {\n      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (NCT + 2), 0);  // DS reads\n", "    if constexpr (false) {\n      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (NCT + 2), 0);\n")]
# the f16x2 form of that kernel with the compiler's own issue order
VARIANTS["h2big_nosched"] = [("conv.hip", "    if constexpr (!H2) {\n      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (NCT + 2), 0);  // DS reads\n", "    if constexpr (!H2 && NCT > 0) {\n      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (NCT + 2), 0);\n"),
                             ("conv.hip", "    } else {\n      // half as many matrix instructions for the same staging:", "    } else if constexpr (false) {\n      // half as many matrix instructions for the same staging:")]
VARIANTS["x3_no192"] = [("conv.hip", "  const bool ok192 = ct != nullptr && kstep == GBK && a->act <= 3 && !x3_big_taps(a);", "  const bool ok192 = false;")]
VARIANTS["ax_oneprod"] = [
    ("attention.hip", "        for (int t = 0; t < (H2 ? 3 : 6); ++t) {", "        for (int t = (H2 ? 2 : 5); t < (H2 ? 3 : 6); ++t) {"),
    ("attention.hip", "          for (int u = 0; u < (H2 ? 3 : 6); ++u) {", "          for (int u = (H2 ? 2 : 5); u < (H2 ? 3 : 6); ++u) {"),
]
VARIANTS["ax_noexp"] = [("attention.hip", "        const float pe = __builtin_amdgcn_exp2f(sacc[r] - m_sub);", "        const float pe = sacc[r] - m_sub;")]
VARIANTS["ax_nosplitp"] = [
    ("attention.hip", "          else az_split3(sacc[8 * s2 + 2 * j], sacc[8 * s2 + 2 * j + 1], p3[0][j], p3[1][j], p3[2][j]);",
     "          else p3[0][j] = p3[1][j] = p3[2][j] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sacc[8 * s2 + 2 * j + 1]), __builtin_bit_cast(unsigned, sacc[8 * s2 + 2 * j]), 0x07060302u);"),
]
# (ax_novt -- the scattered V^T store of round 4 -- went with that code: V is staged row-major and consumed through ds_read_b64_tr_b16)

# ---- gate for a 128-cout x 32-tile block of the x3 Winograd kernel: timing only, WRONG results.  The producer side
#      (gather, staging, B^T d B, V stores) runs in the EVEN cout blocks only (= once per two cout blocks, what a 128-cout block would
#      do); the consumer side reads and splits ONE tile half and feeds it to both (= a V fragment serving four cout quarters).  Not in
#      the gate: the filter fragment loads per MFMA double in that form (tools/ablate.py wx3_nou prices them).
_G_PROD = [
    ("wino_x3.hip", "  const int cb = (rect - rrow * rcols) * p.gc + rin_c;\n", "  const int cb = (rect - rrow * rcols) * p.gc + rin_c;\n  const bool prod = !(cb & 1);\n"),
    ("wino_x3.hip", "    if (m < 5 || m < ndma) {  // (uniform; a block always has at least 520 slots = 4.06 pieces per thread)", "    if (prod && (m < 5 || m < ndma)) {"),
    ("wino_x3.hip", "    if (m < 5 || m < ndma) *reinterpret_cast<float4*>(smem + X_STAGE + ((m * 512 + tid) >> 2) * X_SLOT + (tid & 3) * 16) = gq[m];", "    if (prod && (m < 5 || m < ndma)) *reinterpret_cast<float4*>(smem + X_STAGE + ((m * 512 + tid) >> 2) * X_SLOT + (tid & 3) * 16) = gq[m];"),
    ("wino_x3.hip", "  auto patch_rows = [&](int kt, int r0, int r1) __attribute__((always_inline)) {  // rows [r0, r1) of the staged patch -> rv\n", "  auto patch_rows = [&](int kt, int r0, int r1) __attribute__((always_inline)) {\n    if (!prod) return;\n"),
    ("wino_x3.hip", "    const f32x2 d0 = rv[c], d1 = rv[4 + c], d2 = rv[8 + c], d3 = rv[12 + c];\n", "    if (!prod) return;\n    const f32x2 d0 = rv[c], d1 = rv[4 + c], d2 = rv[8 + c], d3 = rv[12 + c];\n"),
    ("wino_x3.hip", "    const f32x2 u0 = rv[4 * xi], u1 = rv[4 * xi + 1], u2 = rv[4 * xi + 2], u3 = rv[4 * xi + 3];\n", "    if (!prod) return;\n    const f32x2 u0 = rv[4 * xi], u1 = rv[4 * xi + 1], u2 = rv[4 * xi + 2], u3 = rv[4 * xi + 3];\n"),
    ("wino_x3.hip", "    if constexpr (AFF != 0) {\n#pragma unroll\n      for (int i = 0; i < 16; ++i) {\n        f32x2 v = rv[i] * af_sc + af_sh;", "    if constexpr (AFF != 0) {\n      if (!prod) return;\n#pragma unroll\n      for (int i = 0; i < 16; ++i) {\n        f32x2 v = rv[i] * af_sc + af_sh;"),
]
_G_CONS = [
    ("wino_x3.hip", "    const char* vb = smem + hs * X_HALF + fragB + th * (32 * X_ROW);\n", "    if (th == 1) return;\n    const char* vb = smem + hs * X_HALF + fragB + th * (32 * X_ROW);\n"),
    ("wino_x3.hip", "  auto piece = [&](int th, int pl) __attribute__((always_inline)) {  // piece pl = the high halves of the current remainders\n",
     "  auto piece = [&](int th, int pl) __attribute__((always_inline)) {\n    if (th == 1) { for (int j = 0; j < 4; ++j) fw[1][pl][j] = fw[0][pl][j]; return; }\n"),
    ("wino_x3.hip", "  auto remainder = [&](int th, int j0, int j1) __attribute__((always_inline)) {  // x -= its high 16 bits (exact), values 2 j0 .. 2 j1 - 1\n",
     "  auto remainder = [&](int th, int j0, int j1) __attribute__((always_inline)) {\n    if (th == 1) return;\n"),
]
# half of the filter-fragment loads (cout half 1 reuses cout half 0's fragments: operands stay random, unlike wx3_nou): its time
# change x 2 = what the filter stream costs the kernel
VARIANTS["wx3_halfu"] = [("wino_x3.hip", "    for (int ch = 0; ch < 2; ++ch) ua[ch][pl] = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff));",
                          "    for (int ch = 0; ch < 1; ++ch) ua[ch][pl] = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff));\n    ua[1][pl] = ua[0][pl];")]
# ---- gate for F(4x4,3x3) on the split operands (tools/f4_gate.py): timing only, WRONG results.  Launched on a map of
#      0.75 x 0.75 the size, the shipped F(2x2) stream executes exactly the (tile x frequency) work of F(4x4) on the full map: 36 / 16
#      frequencies x 1 / 4 of the tiles = 0.5625 of the matrix instructions, V values (patch reads, transforms, V stores, fragment reads,
#      splits).  What F(4x4) pays on top PER V VALUE is in this variant: the 6 x 6 transform's 4.0 packed operations per value instead
#      of 2.0 (one more packed add behind every add of B^T d and of the nu side), and TWICE the filter fragment bytes per matrix
#      instruction (a 64-cout x 32-tile block: the 295 KB of accumulators of 36 frequencies allow no more; second load 48 KB further
#      on, merged through an opaque zero so that it cannot be dropped).
VARIANTS["wx3_f4proxy"] = [
    ("wino_x3.hip", "  f32x2 rv[16];  // raw 4x4 patch of the channel pair (index = patch row * 4 + column), then B^T d in place\n",
     "  f32x2 rv[16];\n  float zq;\n  asm volatile(\"v_mov_b32 %0, 0\" : \"=v\"(zq));\n  const f32x2 z2 = {zq, zq};\n  const unsigned zu = __builtin_bit_cast(unsigned, zq);\n"),
    ("wino_x3.hip", "    rv[c] = d0 - d2;\n    rv[4 + c] = d1 + d2;\n    rv[8 + c] = d2 - d1;\n    rv[12 + c] = d1 - d3;\n",
     "    rv[c] = (d0 - d2) + z2;\n    rv[4 + c] = (d1 + d2) + z2;\n    rv[8 + c] = (d2 - d1) + z2;\n    rv[12 + c] = (d1 - d3) + z2;\n"),
    ("wino_x3.hip", "    *reinterpret_cast<f32x2*>(dst) = hs == 0 ? u0 - u2 : u2 - u1;\n    *reinterpret_cast<f32x2*>(dst + X_FREQ) = hs == 0 ? u1 + u2 : u1 - u3;\n",
     "    *reinterpret_cast<f32x2*>(dst) = (hs == 0 ? u0 - u2 : u2 - u1) + z2;\n    *reinterpret_cast<f32x2*>(dst + X_FREQ) = (hs == 0 ? u1 + u2 : u1 - u3) + z2;\n"),
    ("wino_x3.hip", "    for (int ch = 0; ch < 2; ++ch) ua[ch][pl] = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff));",
     "    for (int ch = 0; ch < 2; ++ch) {\n      uint4 r0 = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff));\n"
     "      const uint4 r1 = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff + 49152u));\n"
     "      r0.x |= r1.x & zu; r0.y |= r1.y & zu; r0.z |= r1.z & zu; r0.w |= r1.w & zu;\n      ua[ch][pl] = r0;\n    }"),
]
# the same with only one of the two surcharges (which one costs what)
VARIANTS["wx3_f4proxy_adds"] = VARIANTS["wx3_f4proxy"][:3]
VARIANTS["wx3_f4proxy_u"] = [VARIANTS["wx3_f4proxy"][0], VARIANTS["wx3_f4proxy"][3]]
VARIANTS["wx3_g_halfprod"] = _G_PROD
VARIANTS["wx3_g_halfcons"] = _G_CONS
VARIANTS["wx3_g_128x32"] = _G_PROD + _G_CONS

# ---- streaming (HBM-bound) kernels and non-temporal accesses (RESULTS STAY CORRECT: only the cache policy changes); measured by
#      tools/stream_ab.py -> profiles/r05_stream_nt_ab.txt.  The transition kernels ship with nt loads and stores (common.h:
#      az_ld_stream / az_st_stream); az_affine_act_f32 ships plain.
VARIANTS["tr_plain"] = [  # the transition kernels as before: plain loads and stores
    ("common.h", "  const az_f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const az_f32x4_t*>(p));", "  const az_f32x4_t v = *reinterpret_cast<const az_f32x4_t*>(p);"),
    ("common.h", "  __builtin_nontemporal_store(v, reinterpret_cast<az_f32x4_t*>(p));", "  *reinterpret_cast<az_f32x4_t*>(p) = v;")]
VARIANTS["tr_ntld"] = VARIANTS["tr_plain"][1:]  # nt loads, plain stores
VARIANTS["tr_ntst"] = VARIANTS["tr_plain"][:1]  # plain loads, nt stores
VARIANTS["tr_un8"] = [("transition.hip", "constexpr int TF_UN = 4;", "constexpr int TF_UN = 8;")]  # 8 float4 per thread and stream in flight
VARIANTS["tr_un2"] = [("transition.hip", "constexpr int TF_UN = 4;", "constexpr int TF_UN = 2;")]
VARIANTS["tr_grid2k"] = [("transition.hip", "  const int64_t gcap = 16384 / gy < 1 ? 1 : 16384 / gy;", "  const int64_t gcap = 2048 / gy < 1 ? 1 : 2048 / gy;")]  # 8 resident workgroups per CU, grid-stride
# the matrix kernels' output stores (epilogue_batch_nhwc) ship as non-temporal stores for outputs of 128 MiB and more:
# never / always the hint
VARIANTS["epi_plainst"] = [("conv_shared.h", "        if (stream_out) az_st_stream(d, f);\n        else *reinterpret_cast<float4*>(d) = f;", "        *reinterpret_cast<float4*>(d) = f;")]
VARIANTS["epi_ntall"] = [("conv_shared.h", "        if (stream_out) az_st_stream(d, f);\n        else *reinterpret_cast<float4*>(d) = f;", "        az_st_stream(d, f);")]
# the x3 Winograd kernel's filter stream with the non-temporal hint (aux = 2): the small-map layers read every filter byte once or twice
VARIANTS["wx3_unt"] = [("wino_x3.hip", "ua[ch][pl] = __builtin_bit_cast(uint4, buf_ld4(rw, u_lane + (unsigned)((ch * 3 + pl) * 1024), soff));",
                        "ua[ch][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(u_lane + (unsigned)((ch * 3 + pl) * 1024)), (int)soff, 2));")]
# further candidates for the hint (one line each): the row norms' output, the stem's output, the epilogue's residual reads
VARIANTS["rn_ntst"] = [("norm.hip", "          *reinterpret_cast<float4*>(yr + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);", "          az_st_stream(yr + c, make_float4(o4[0], o4[1], o4[2], o4[3]));")]
VARIANTS["stem_ntst"] = [("conv.hip", "        *reinterpret_cast<float4*>(a.dst + (((int64_t)b * H + oh) * W + ow) * a.cout_s + q * 4) = o;", "        az_st_stream(a.dst + (((int64_t)b * H + oh) * W + ow) * a.cout_s + q * 4, o);")]
VARIANTS["res_ntld"] = [("conv_shared.h", "      if constexpr (RES == 1) r[i] = ld4_io<IO>(a.res, (int64_t)n[i] * a.cout_s + co);", "      if constexpr (RES == 1) r[i] = az_ld_stream(a.res + (int64_t)n[i] * a.cout_s + co);")]
_AF_LD = [("norm.hip", "  if (x1 == nullptr) return ld4_io<IO>(x, pix * cs + c);", "  if (x1 == nullptr) return az_ld_stream(x + pix * cs + c);")]
_AF_ST = [("norm.hip", "    for (int u = 0; u < UN; ++u) st4_io<IO>(y, yo + (int64_t)(p + u * pstride) * cs, apply(v[u]));",
           "    for (int u = 0; u < UN; ++u) az_st_stream(y + yo + (int64_t)(p + u * pstride) * cs, apply(v[u]));")]
VARIANTS["aff_ntld"] = _AF_LD
VARIANTS["aff_ntst"] = _AF_ST
VARIANTS["aff_nt"] = _AF_LD + _AF_ST

# ---- f16x2 form of the x3 Winograd kernel (wino_x3.hip, H2): the low piece by v_fma_mixlo / mixhi_f16 (fp32 fma with an f16 source,
#      rounded to f16 in the same instruction: 2 instead of 4 vector instructions per pair) -- RESULTS STAY CORRECT
VARIANTS["wx3h_mix"] = [("wino_x3.hip",
    "      const h2v l = {(_Float16)__builtin_fmaf((float)hh.x, -2048.f, xf[th][2 * j]), (_Float16)__builtin_fmaf((float)hh.y, -2048.f, xf[th][2 * j + 1])};\n      fw[th][1][j] = __builtin_bit_cast(unsigned, l);\n",
    "      (void)hh;\n      unsigned lw;\n"
    "      asm(\"v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\" : \"=v\"(lw) : \"v\"(fw[th][0][j]), \"s\"(-2048.f), \"v\"(xf[th][2 * j]));\n"
    "      asm(\"v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\" : \"+v\"(lw) : \"v\"(fw[th][0][j]), \"s\"(-2048.f), \"v\"(xf[th][2 * j + 1]));\n"
    "      fw[th][1][j] = lw;\n")]

# ---- f16x2 form of the x3 Winograd kernel: slot-order experiments (RESULTS STAY CORRECT)
_H2_P0_OLD = ("        mf(hs, 0); h_piece(1); XS_FENCE;\n"
              "        mf(hs, 1); nu_store(ob, 0); nu_store(ob, 1); XS_FENCE;\n"
              "        mf(hs, 2); nu_store(ob, 2); nu_store(ob, 3); gl(ktn, 0); gl(ktn, 1); gl(ktn, 2); XS_FENCE;  // (rv is free from here)\n"
              "        mf(hs, 3); gl(ktn, 3); gl(ktn, 4); gl(ktn, 5); load_u(ktu, ob, 1); XS_FENCE;  // (a1 is free: the next phase's a1)\n"
              "        mf(hs, 4); l_piece(0); XS_FENCE;\n"
              "        mf(hs, 5); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;\n"
              "        mf(hs, 6); XS_FENCE;\n"
              "        mf(hs, 7); load_u(ktu, ob, 0); XS_FENCE;  // (a0 is free)\n"
              "        mf(hs, 8); gs(0); XS_FENCE;\n"
              "        mf(hs, 9); gs(1); gs(2); XS_FENCE;\n"
              "        mf(hs, 10); gs(3); gs(4); XS_FENCE;\n"
              "        mf(hs, 11); gs(5);\n")
_H2_P1_OLD = ("        mf(hs, 0); h_piece(1); affine_load(ktn); XS_FENCE;\n"
              "        mf(hs, 1); patch_rows(ktn, 0, 2); XS_FENCE;\n"
              "        mf(hs, 2); patch_rows(ktn, 2, 4); XS_FENCE;\n"
              "        mf(hs, 3); l_piece(0); load_u(ktu, ob, 1); XS_FENCE;\n"
              "        mf(hs, 4); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;\n"
              "        affine();\n"
              "        mf(hs, 5); row_transform(0); row_transform(1); XS_FENCE;\n"
              "        mf(hs, 6); row_transform(2); row_transform(3); XS_FENCE;\n"
              "        mf(hs, 7); load_u(ktu, ob, 0); XS_FENCE;\n"
              "        mf(hs, 8); nu_store(ob, 0); XS_FENCE;\n"
              "        mf(hs, 9); nu_store(ob, 1); XS_FENCE;\n"
              "        mf(hs, 10); nu_store(ob, 2); XS_FENCE;\n"
              "        mf(hs, 11); nu_store(ob, 3);\n")
# (a) two products per slot: half the fences, the compiler orders inside a slot
VARIANTS["wx3h_2mf"] = [
    ("wino_x3.hip", _H2_P0_OLD,
     "        mf(hs, 0); mf(hs, 1); h_piece(1); nu_store(ob, 0); nu_store(ob, 1); XS_FENCE;\n"
     "        mf(hs, 2); mf(hs, 3); nu_store(ob, 2); nu_store(ob, 3); gl(ktn, 0); gl(ktn, 1); gl(ktn, 2); load_u(ktu, ob, 1); XS_FENCE;\n"
     "        mf(hs, 4); mf(hs, 5); gl(ktn, 3); gl(ktn, 4); gl(ktn, 5); l_piece(0); XS_FENCE;\n"
     "        mf(hs, 6); mf(hs, 7); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); load_u(ktu, ob, 0); XS_FENCE;\n"
     "        mf(hs, 8); mf(hs, 9); gs(0); gs(1); gs(2); gs(3); XS_FENCE;\n"
     "        mf(hs, 10); mf(hs, 11); gs(4); gs(5);\n"),
    ("wino_x3.hip", _H2_P1_OLD,
     "        mf(hs, 0); mf(hs, 1); h_piece(1); affine_load(ktn); patch_rows(ktn, 0, 2); XS_FENCE;\n"
     "        mf(hs, 2); mf(hs, 3); patch_rows(ktn, 2, 4); l_piece(0); load_u(ktu, ob, 1); XS_FENCE;\n"
     "        mf(hs, 4); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;\n"
     "        affine();\n"
     "        mf(hs, 5); row_transform(0); row_transform(1); XS_FENCE;\n"
     "        mf(hs, 6); mf(hs, 7); row_transform(2); row_transform(3); load_u(ktu, ob, 0); XS_FENCE;\n"
     "        mf(hs, 8); mf(hs, 9); nu_store(ob, 0); nu_store(ob, 1); XS_FENCE;\n"
     "        mf(hs, 10); mf(hs, 11); nu_store(ob, 2); nu_store(ob, 3);\n")]
# (b) the order of the first f16x2 builds: staging loads one slot later, their LDS stores two to a slot (the tree's order measured 2.4 - 3 % faster)
VARIANTS["wx3h_lategl"] = [
    ("wino_x3.hip", _H2_P0_OLD,
     "        mf(hs, 0); h_piece(1); XS_FENCE;\n"
     "        mf(hs, 1); nu_store(ob, 0); nu_store(ob, 1); XS_FENCE;\n"
     "        mf(hs, 2); nu_store(ob, 2); nu_store(ob, 3); XS_FENCE;\n"
     "        mf(hs, 3); gl(ktn, 0); gl(ktn, 1); gl(ktn, 2); load_u(ktu, ob, 1); XS_FENCE;\n"
     "        mf(hs, 4); gl(ktn, 3); gl(ktn, 4); gl(ktn, 5); XS_FENCE;\n"
     "        mf(hs, 5); l_piece(0); XS_FENCE;\n"
     "        mf(hs, 6); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;\n"
     "        mf(hs, 7); load_u(ktu, ob, 0); XS_FENCE;\n"
     "        mf(hs, 8); gs(0); gs(1); XS_FENCE;\n"
     "        mf(hs, 9); gs(2); gs(3); XS_FENCE;\n"
     "        mf(hs, 10); gs(4); gs(5); XS_FENCE;\n"
     "        mf(hs, 11);\n")]
# (b2) the staging loads one slot earlier still (behind the first two V stores)
VARIANTS["wx3h_gl1"] = [
    ("wino_x3.hip", _H2_P0_OLD,
     "        mf(hs, 0); h_piece(1); XS_FENCE;\n"
     "        mf(hs, 1); nu_store(ob, 0); nu_store(ob, 1); nu_store(ob, 2); nu_store(ob, 3); XS_FENCE;\n"
     "        mf(hs, 2); gl(ktn, 0); gl(ktn, 1); gl(ktn, 2); gl(ktn, 3); gl(ktn, 4); gl(ktn, 5); XS_FENCE;\n"
     "        mf(hs, 3); l_piece(0); load_u(ktu, ob, 1); XS_FENCE;\n"
     "        mf(hs, 4); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;\n"
     "        mf(hs, 5); XS_FENCE;\n"
     "        mf(hs, 6); XS_FENCE;\n"
     "        mf(hs, 7); load_u(ktu, ob, 0); XS_FENCE;\n"
     "        mf(hs, 8); gs(0); XS_FENCE;\n"
     "        mf(hs, 9); gs(1); gs(2); XS_FENCE;\n"
     "        mf(hs, 10); gs(3); gs(4); XS_FENCE;\n"
     "        mf(hs, 11); gs(5);\n")]
# (b3) phase 1: the low pieces first, the patch reads behind them
VARIANTS["wx3h_p1"] = [
    ("wino_x3.hip", _H2_P1_OLD,
     "        mf(hs, 0); h_piece(1); affine_load(ktn); XS_FENCE;\n"
     "        mf(hs, 1); l_piece(0); XS_FENCE;\n"
     "        mf(hs, 2); l_piece(1); if constexpr (H2_DERIVE) derive_u2(); XS_FENCE;\n"
     "        mf(hs, 3); patch_rows(ktn, 0, 2); load_u(ktu, ob, 1); XS_FENCE;\n"
     "        mf(hs, 4); patch_rows(ktn, 2, 4); XS_FENCE;\n"
     "        affine();\n"
     "        mf(hs, 5); row_transform(0); row_transform(1); XS_FENCE;\n"
     "        mf(hs, 6); row_transform(2); row_transform(3); XS_FENCE;\n"
     "        mf(hs, 7); load_u(ktu, ob, 0); XS_FENCE;\n"
     "        mf(hs, 8); nu_store(ob, 0); XS_FENCE;\n"
     "        mf(hs, 9); nu_store(ob, 1); XS_FENCE;\n"
     "        mf(hs, 10); nu_store(ob, 2); XS_FENCE;\n"
     "        mf(hs, 11); nu_store(ob, 3);\n")]
# (b4) phase 1 without fences (the compiler's order there, the pinned order in phase 0)
VARIANTS["wx3h_p1nofence"] = [
    ("wino_x3.hip", _H2_P1_OLD, _H2_P1_OLD.replace("XS_FENCE;", ""))]
# (c) no fences at all in the f16x2 phases: the compiler's own order
VARIANTS["wx3h_nofence"] = [
    ("wino_x3.hip", "    if constexpr (H2) {\n      // 12 products per phase; the producer side is the same work as below, two slots' worth per slot\n",
     "    if constexpr (H2) {\n#undef XS_FENCE\n#define XS_FENCE (void)0\n"),
    ("wino_x3.hip", "      if constexpr (!H2_DERIVE) load_u(ktu, ob, 2);\n    } else {\n",
     "      if constexpr (!H2_DERIVE) load_u(ktu, ob, 2);\n#undef XS_FENCE\n#define XS_FENCE __builtin_amdgcn_sched_barrier(0)\n    } else {\n")]
# ---- f16x2 form of the 256 x 256 GEMM: issue-pattern experiments
_H2G_OLD = "        if (k < NM - 6) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three vector instructions\n"
VARIANTS["h2big_valu2"] = [("conv.hip", _H2G_OLD, "        if (k < NM - 4) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);\n")]
VARIANTS["h2big_valu4"] = [("conv.hip", _H2G_OLD, "        if (k < NM - 10) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);\n")]

# ---- x3 Winograd kernel: staged pixel slots at a stride of 80 B instead of 64 B (measured: no gain, more bank conflicts -- profiles/r06_wx3_slot_ab.txt)
VARIANTS["wx3_slot80"] = [("wino_x3.hip", "constexpr int X_SLOT = 64;", "constexpr int X_SLOT = 80;")]

# ---- f16x2 attention, 256-query workgroups at head_dim <= 64: four waves per SIMD (<= 128 registers: two resident workgroups per CU, so
#      that one's load / stage / barrier skeleton hides under the other's products) -- the compiler spills 31 registers to get there
VARIANTS["att_h2_2wg"] = [("attention.hip", "__global__ __launch_bounds__(64 * NW) void attention_x3_kernel(AzAttnArgs a) {",
                           "__global__ __launch_bounds__(64 * NW, (H2 && NW == 8 && D <= 64) ? 4 : 1) void attention_x3_kernel(AzAttnArgs a) {")]


def build(name: str, patches=None, regen_env=None, head_files=None) -> str:
    r"""`patches`: a substitution list instead of VARIANTS[name]; `regen_env`: generator overrides (KL_* / KG_*) -- the copy's
    wino_kloop.inc / igemm_kloop.inc are regenerated with them (tools/kloop_variant.py)."""
    out_dir = os.path.join(B.HERE, "_ab")
    src_dir = os.path.join(out_dir, "src_" + name, "azula_amd", "csrc")
    shutil.rmtree(os.path.join(out_dir, "src_" + name), ignore_errors=True)
    os.makedirs(src_dir)
    os.makedirs(os.path.join(out_dir, "src_" + name, "include"))
    shutil.copy(os.path.join(ROOT, "include", "azula_amd.h"), os.path.join(out_dir, "src_" + name, "include"))
    for f in os.listdir(B.HERE):
        if f.endswith((".hip", ".h", ".inc")):
            shutil.copy(os.path.join(B.HERE, f), src_dir)
    touched = set()
    if regen_env is not None:
        env = dict(os.environ, AZ_KLOOP_AB="1", **regen_env)
        for gen, inc in (("gen_wino_kloop.py", "wino_kloop.inc"), ("gen_igemm_kloop.py", "igemm_kloop.inc")):
            subprocess.run([sys.executable, os.path.join(B.HERE, gen), "--out", os.path.join(src_dir, inc)], check=True, env=env, stdout=subprocess.DEVNULL)
        touched.add("conv.hip")
    for f in (head_files or ()):
        open(os.path.join(src_dir, f), "w").write(subprocess.run(["git", "show", "HEAD:azula_amd/csrc/" + f], check=True, capture_output=True, text=True, cwd=ROOT).stdout)
        touched.add(f)
    for f, old, new in (() if head_files else VARIANTS[name] if patches is None else patches):
        path = os.path.join(src_dir, f)
        text = open(path).read()
        assert text.count(old) == 1, f"{name}: pattern occurs {text.count(old)} times in {f}: {old[:60]!r}"
        open(path, "w").write(text.replace(old, new))
        touched.add(f)
    if any(f.endswith((".h", ".inc")) for f in touched):  # a patched header: every translation unit is rebuilt from the copy
        touched |= set(B.SOURCES)
    objs = []
    procs = []
    for s in B.SOURCES:
        if s in touched:
            obj = os.path.join(src_dir, s.replace(".hip", ".o"))
            procs.append(subprocess.Popen([B.hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(s, []), "-x", "hip", "-c", os.path.join(src_dir, s), "-o", obj]))
        else:  # untouched translation units: the tree's own objects
            obj = os.path.join(B.OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
    assert all(p.wait() == 0 for p in procs)
    lib = os.path.join(out_dir, f"libazula_amd_{name}.so")
    subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    shutil.rmtree(os.path.join(out_dir, "src_" + name), ignore_errors=True)
    return lib


if __name__ == "__main__":
    if sys.argv[1:] == ["--list"]:
        print("\n".join(VARIANTS))
    elif sys.argv[1:2] == ["--head"]:
        B.build()
        print(build("head", head_files=sys.argv[2:]))
    else:
        B.build()
        for n in sys.argv[1:]:
            print(build(n))
