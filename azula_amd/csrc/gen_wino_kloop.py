r"""Generates ``wino_kloop.inc``: the K loop of ``conv_winograd_kernel`` as two hand-scheduled gfx950 instruction streams
(one per loader role), wrapped as string-literal macros for ``asm volatile``.

Why: the C++ loop is [loads][32 MFMAs][transform + LDS stores][barrier]: after every barrier the matrix pipe idles for the
fragment-read latency, both waves of a SIMD then leave the MFMA phase together and stage with an idle pipe
(5.6 k cycles per 8-channel stage against 4.1 k of MFMA issue).  ``profiles/r03_mfma_valu_overlap.txt`` shows what
may sit beside ``v_mfma_f32_32x32x2_f32`` for free (<= 4 ds_read_b128, <= 2 ds_write_b64, 1 buffer load, any SALU per
64-cycle MFMA) and what may not (every VALU instruction costs its own issue time).  So the stream here is software
pipelined across the barrier:

    iteration t:   s_barrier                                   (S(t) visible; every R(t-1) done)
                   R(t)[0], R(t)[1]                            fragment reads of frequency groups 0, 1
                   M(t-1)[6], M(t-1)[7]                        8 MFMAs whose fragments were read BEFORE the barrier
                   M(t)[0..5] with R(t)[2..7] two groups ahead
                   beside the first 18 MFMAs: S(t+1) (wait for L(t+1), input transform, LDS stores into the other
                   buffer) and L(t+2) (global loads, one stage of latency cover), one memory instruction per MFMA gap

The matrix pipe therefore always has MFMAs queued when a wave reaches the barrier and right after it.  Register use is
fixed (v160..v255, s91..s99: declared as clobbers); accumulators and the few scalar inputs are asm operands.
Environment overrides KL_* (ablations, scalar adds, LDS-DMA filter path, padding) exist for tools/kloop_variant.py A/B builds
only and are ignored unless AZ_KLOOP_AB=1.

    python azula_amd/csrc/gen_wino_kloop.py              # rewrites wino_kloop.inc next to this file (developer command)
    python azula_amd/csrc/gen_wino_kloop.py --out PATH   # writes a variant elsewhere (A/B builds)
    python azula_amd/csrc/gen_wino_kloop.py --check      # exit 1 if the committed .inc is not what the generator emits
"""
from __future__ import annotations

import os

HERE = os.path.dirname(os.path.abspath(__file__))
# The KL_* overrides below (ablations, scalar adds, LDS-DMA filter path, padding) are honoured ONLY when AZ_KLOOP_AB=1 is set
# as well (tools/kloop_variant.py does): a user's environment must never change the shipped instruction stream.
_AB = os.environ if os.environ.get("AZ_KLOOP_AB") == "1" else {}

TB = 160
SP, TP = 156, 158   # input affine (AzConvArgs.in_affine): scale / shift of the thread's channel pair for the stage being transformed
RV0, VO0, X0, Y0 = 192, 224, 240, 248
V_FA, V_FB, V_ST, V_OOB = 252, 253, 254, 255
S_FIRST, S_RS, S_SOFF, S_KT, S_CNT, S_TMP = 91, 92, 96, 97, 98, 99  # (s100 / s101 are reserved on this target)
S_AF, S_AFT, S_AFS, S_ALLV = 84, 88, 89, 90  # in_affine: descriptor s[84:87], byte distance scale -> shift (0 = no affine), soffset temp,
#                                             1 = this wave's patches have no padding position
GROUP_BYTES = 2048  # one frequency of a stage: 64 rows x 8 floats x 4 B
BUF_XOR = 0x10000   # the two 64 KB stages


def fa(s, j=None):
    return f"v[{TB + 8 * s}:{TB + 8 * s + 3}]" if j is None else f"v{TB + 8 * s + j}"


def fb(s, j=None):
    return f"v[{TB + 8 * s + 4}:{TB + 8 * s + 7}]" if j is None else f"v{TB + 8 * s + 4 + j}"


def pair(r):
    return f"v[{r}:{r + 1}]"


def quad(r):
    return f"v[{r}:{r + 3}]"


class Stream:
    r"""Ordered instruction list with the bookkeeping for `s_waitcnt lgkmcnt(N)`: LDS operations retire in order, so
    waiting for the reads of fragment set s means allowing as many outstanding operations as were issued after them."""

    def __init__(self):
        self.lines: list[str] = []
        self.lds_seq = 0                      # LDS operations issued so far
        self.set_done_at: dict[int, int] = {}  # fragment set -> lds_seq right after its second read

    def emit(self, text: str):
        self.lines.append(text)

    def lds(self, text: str):
        self.lines.append(text)
        self.lds_seq += 1

    def read_set(self, s: int, group: int):
        if "afrag" not in ABL:  # (timing ablation: the filter fragments never read -- what keeping U out of LDS would save)
            self.lds(f"ds_read_b128 {fa(s)}, v{V_FA} offset:{group * GROUP_BYTES}")
        self.lds(f"ds_read_b128 {fb(s)}, v{V_FB} offset:{group * GROUP_BYTES}")
        self.set_done_at[s] = self.lds_seq

    def need_set(self, s: int):
        n = self.lds_seq - self.set_done_at[s]
        self.emit(f"s_waitcnt lgkmcnt({min(n, 15)})")

    def drain(self):
        self.emit("s_waitcnt lgkmcnt(0)")
        self.set_done_at = {k: 0 for k in self.set_done_at}
        self.lds_seq = 0

    def mfma(self, acc: int, s: int, j: int):
        self.emit(f"v_mfma_f32_32x32x2_f32 %{acc}, {fa(s, j)}, {fb(s, j)}, %{acc}")


# MFMA gaps of the transform's vector work: pass 1 per patch column, pass 2 per row (a burst pays one pipe switch, ~10 cycles)
# (measured on the 256^2 / 64^2 layers, same box: pass 1 in two bursts + pass 2 a row per store gap 1256 / 336 us; pass 1 in one
# burst, rows 0-1 and rows 2-3 in one each 1229 - 1245 / 327 - 331 us)
P1_GAPS = [int(v) for v in _AB.get("KL_P1", "0,0,0,0").split(",")]
P2_GAPS = [int(v) for v in _AB.get("KL_P2", "1,1,5,5").split(",")]
SCALAR_ADD = _AB.get("KL_SCALAR_ADD", "0") == "1"  # A/B: two v_add / v_sub instead of one v_pk_add_f32


def pk(dst, a, b, sub=False):
    r"""dst = a +/- b on a register pair (text; two lines joined by a newline in the scalar form)."""
    if not SCALAR_ADD:
        return f"v_pk_add_f32 {pair(dst)}, {pair(a)}, {pair(b)}" + (" neg_lo:[0,1] neg_hi:[0,1]" if sub else "")
    op = "v_sub_f32" if sub else "v_add_f32"
    return f"{op} v{dst}, v{a}, v{b}\\n{op} v{dst + 1}, v{a + 1}, v{b + 1}"


# ------------------------------------------------------------------------------------------------ V role pieces
def v_pass1_col(c, rv0=RV0):
    d0, d1, d2, d3 = rv0 + 2 * c, rv0 + 2 * (4 + c), rv0 + 2 * (8 + c), rv0 + 2 * (12 + c)
    x = X0 + 2 * c
    return [
        pk(d0, d0, d2, True),   # o0 = d0 - d2
        pk(d3, d1, d3, True),   # o3 = d1 - d3
        pk(x, d1, d2),          # o1 = d1 + d2 -> X[c]
        pk(d2, d2, d1, True),   # o2 = d2 - d1
    ]


def v_row_regs(xi, rv0=RV0):
    if xi == 1:
        return [X0 + 2 * c for c in range(4)]
    return [rv0 + 2 * (4 * xi + c) for c in range(4)]


def v_pass2_row(xi, rv0=RV0):
    r"""Row xi of the second transform pass, in place (out1 in a Y temp), and its two LDS stores."""
    u0, u1, u2, u3 = v_row_regs(xi, rv0)
    y = Y0 + 2 * (xi & 1)
    valu = [
        ("valu", pk(u0, u0, u2, True)),  # out0 = u0 - u2
        ("valu", pk(y, u1, u2)),         # out1 = u1 + u2
        ("valu", pk(u2, u2, u1, True)),  # out2 = u2 - u1
        ("valu", pk(u3, u1, u3, True)),  # out3 = u1 - u3
    ]
    wa = ("lds", f"ds_write2st64_b64 v{V_ST}, {pair(u0)}, {pair(y)} offset0:{16 * xi} offset1:{16 * xi + 4}")
    wb = ("lds", f"ds_write2st64_b64 v{V_ST}, {pair(u2)}, {pair(u3)} offset0:{16 * xi + 8} offset1:{16 * xi + 12}")
    return valu, wa, wb


def v_load(i, rv0=RV0):
    return ("vmem", f"buffer_load_dwordx2 {pair(rv0 + 2 * i)}, v{VO0 + i}, s[{S_RS}:{S_RS + 3}], s{S_SOFF} offen")


V_OPS = dict(fragA=8, fragB=9, st=10, ldsA=11, ldsB=12, rs0=13, rs1=17, kt_begin=21, kt_end=22, kt_switch=23, soff0=24,
             tail0=25, tail1=26, af_voff=27, af_lo=28, af_hi=29, af_delta=30)


AFFINE = False   # set by gen_role: the V stream with AzConvArgs.in_affine (a second asm statement; the plain stream has no trace of it --
#                  a TAKEN scalar branch costs the wave ~50 cycles, four skipped blocks per stage were 2 % of the kernel)
OOL: list = []   # out-of-line blocks of the stream being generated (emitted behind its last instruction)


S_MASK = 52      # in_affine: s[52:83] = 16 lane masks, position i valid (not padding) for the lane


def v_affine_col(c, tag, rv0=RV0):
    r"""In-place y = x * scale + shift on column c of the raw patch (4 positions x a channel pair).  Padding positions (patch
    offset = OOB: negative) must stay zero -- the convolution pads the NORMALISED tensor: their lanes are switched off for the
    FMA (the load left 0 there) by scalar moves of precomputed lane masks into EXEC; the VALU instructions of this stream are
    not hidden behind the MFMAs (each costs its issue time), scalar ones are."""
    if not AFFINE:
        return []
    L = []
    for r in range(4):
        i = 4 * r + c
        d = rv0 + 2 * i
        L.append(("salu", f"s_mov_b64 exec, s[{S_MASK + 2 * i}:{S_MASK + 2 * i + 1}]"))
        L.append(("valu", f"v_pk_fma_f32 {pair(d)}, {pair(d)}, {pair(SP)}, {pair(TP)}"))
    L.append(("salu", "s_mov_b64 exec, -1"))
    return L


def v_affine_loads(tag, back=0):
    r"""scale / shift pairs of the NEXT stage to transform (soffset: the stage's channel offset, `back` bytes behind s_soff)."""
    if not AFFINE:
        return []
    o = V_OPS
    L = [("salu", f"s_sub_u32 s{S_AFS}, s{S_SOFF}, {back}")]
    L.append(("vmem", f"buffer_load_dwordx2 {pair(SP)}, %{o['af_voff']}, s[{S_AF}:{S_AF + 3}], s{S_AFS} offen"))
    L.append(("salu", f"s_add_u32 s{S_AFS}, s{S_AFS}, s{S_AFT}"))
    L.append(("vmem", f"buffer_load_dwordx2 {pair(TP)}, %{o['af_voff']}, s[{S_AF}:{S_AF + 3}], s{S_AFS} offen"))
    return L


OOL_EVENTS = _AB.get("KL_OOL_EVENTS", "1") == "1"  # events out of line: the common path takes no branch


def v_load_events(tag):
    r"""Before the loads of stage s_kt: switch to the second source (descriptor, patch offsets from LDS, channel offset 0)
    and / or mask the channel pairs beyond a source's last (half) chunk.  Both are rare: their code sits behind the stream's
    last instruction, so that the common path falls through untaken branches."""
    o = V_OPS
    sw = [f"s_mov_b32 s{S_RS + w}, %{o['rs1'] + w}" for w in range(4)] + [f"s_mov_b32 s{S_SOFF}, 0"]
    sw += [f"ds_read_b128 {quad(VO0 + 4 * q)}, %{o['ldsB']} offset:{16 * q}" for q in range(4)] + ["s_waitcnt lgkmcnt(0)"]
    tl = ["s_mov_b32 vcc_lo, 0x33333333", "s_mov_b32 vcc_hi, 0x33333333"]
    tl += [f"v_cndmask_b32 v{VO0 + i}, v{V_OOB}, v{VO0 + i}, vcc" for i in range(16)]
    L = []
    if OOL_EVENTS:
        L.append(("salu", f"s_cmp_eq_u32 s{S_KT}, %{o['kt_switch']}"))
        L.append(("salu", f"s_cbranch_scc1 LVsw{tag}_%="))
        L.append(("label", f"LVnsw{tag}_%=:"))
        L.append(("salu", f"s_cmp_eq_u32 s{S_KT}, %{o['tail0']}"))
        L.append(("salu", f"s_cbranch_scc1 LVtl{tag}_%="))
        L.append(("salu", f"s_cmp_eq_u32 s{S_KT}, %{o['tail1']}"))
        L.append(("salu", f"s_cbranch_scc1 LVtl{tag}_%="))
        L.append(("label", f"LVntl{tag}_%=:"))
        OOL.append([f"LVsw{tag}_%=:"] + sw + [f"s_branch LVnsw{tag}_%="])
        OOL.append([f"LVtl{tag}_%=:"] + tl + [f"s_branch LVntl{tag}_%="])
        return L
    L.append(("salu", f"s_cmp_eq_u32 s{S_KT}, %{o['kt_switch']}"))
    L.append(("salu", f"s_cbranch_scc0 LVnsw{tag}_%="))
    L += [("ldsx", t) for t in sw]
    L.append(("label", f"LVnsw{tag}_%=:"))
    L.append(("salu", f"s_cmp_eq_u32 s{S_KT}, %{o['tail0']}"))
    L.append(("salu", f"s_cbranch_scc1 LVtl{tag}_%="))
    L.append(("salu", f"s_cmp_eq_u32 s{S_KT}, %{o['tail1']}"))
    L.append(("salu", f"s_cbranch_scc0 LVntl{tag}_%="))
    L.append(("label", f"LVtl{tag}_%=:"))
    L += [("valu", t) for t in tl]
    L.append(("label", f"LVntl{tag}_%=:"))
    return L


def v_load_done():
    return [("salu", f"s_add_u32 s{S_SOFF}, s{S_SOFF}, 32"), ("salu", f"s_add_u32 s{S_KT}, s{S_KT}, 1")]


def v_extras(S: bool, L: bool, tag: str):
    r"""{MFMA index: [(kind, text)]} -- what is issued behind each of the first MFMAs of an iteration.  VALU work comes in
    bursts (the first vector instruction behind an MFMA costs ~14 cycles, each further one 4-5: profiles/r03_mfma_overlap.txt),
    memory instructions one per gap (one buffer load per 64 cycles per SIMD is what the texture path takes)."""
    ex = {k: [] for k in range(32)}
    if S:
        ex[0].append(("wait", "s_waitcnt vmcnt(0)"))
        for c in range(4):
            ex[P1_GAPS[c]] += v_affine_col(c, tag)
            ex[P1_GAPS[c]] += [("valu", t) for t in v_pass1_col(c)]
        for xi in range(4):  # (a row's stores stay in gaps 2 + 2 xi, 3 + 2 xi; its adds may run earlier, in a larger burst)
            valu, wa, wb = v_pass2_row(xi)
            ex[P2_GAPS[xi]] += valu
            ex[2 + 2 * xi] += [wa]
            ex[3 + 2 * xi] += [wb]
    if L:
        # patch row 1 lives in X after pass 1, so its registers are free first; the other rows follow their stores
        ex[2] = ex[2] + v_load_events(tag)
        order = [4, 5, 6, 7, 0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15]
        for n, i in enumerate(order):
            ex[2 + n].append(v_load(i))
        ex[17] += v_load_done()
        ex[18] += v_affine_loads(tag, back=32)   # (s_soff already points at the stage after the one just requested)
    if S and not XOR_AT_END:
        ex[17].append(("valu", f"v_xor_b32 v{V_ST}, 0x{BUF_XOR:x}, v{V_ST}"))
    return ex


# ------------------------------------------------------------------------------------------------ U role pieces
U_OPS = dict(fragA=8, fragB=9, st=10, voff=11, rw=12, kt_begin=16, kt_end=17, soff0=18, soff_step=19)


UDMA = False  # (round-3 experiment: filter chunk by LDS-DMA; its operand set left conv.hip in round 5 -- measured a tie, DESIGN_HISTORY)
S_UST = S_KT  # (the filter role has no stage counter: its SGPR holds the wave's LDS slot in the buffer to fill next)


def u_dma(i: int, o) -> list:
    r"""16 B per lane of filter chunk piece i straight into LDS: M0 = this wave's 1 KB slot, the per-lane SOURCE offset carries
    the XOR swizzle of the LDS image (the destination of an LDS-DMA is wave-uniform base + 16 x lane)."""
    L = [("salu", f"s_add_u32 m0, s{S_UST}, {4096 * i}")]
    if i:
        L.append(("salu", f"s_add_u32 s{S_TMP}, s{S_SOFF}, {4096 * i}"))
    else:
        L.append(("salu", "s_nop 0"))
    L.append(("vmem", f"buffer_load_dwordx4 %{o['voff']}, s[{S_RS}:{S_RS + 3}], s{S_TMP if i else S_SOFF} offen lds"))
    return L


def u_extras(S: bool, L: bool, tag: str):
    ex = {k: [] for k in range(32)}
    o = U_OPS
    if UDMA:
        if S:  # stage t + 1 into the buffer that became free at this iteration's barrier; landed before the next barrier
            for i in range(8):
                ex[1 + i] += u_dma(i, o)
            ex[8].append(("salu", f"s_add_u32 s{S_SOFF}, s{S_SOFF}, %{o['soff_step']}"))
            ex[8].append(("salu", f"s_xor_b32 s{S_UST}, s{S_UST}, 0x{BUF_XOR:x}"))
            ex[31].append(("wait", "s_waitcnt vmcnt(0)"))
        return ex
    if S:
        ex[0].append(("wait", "s_waitcnt vmcnt(0)"))
        for i in range(8):
            ex[i].append(("lds", f"ds_write_b128 v{V_ST}, {quad(RV0 + 4 * i)} offset:{4096 * i}"))
    if L:
        for i in range(8):
            k = i + 1
            if i:
                ex[k].append(("salu", f"s_add_u32 s{S_TMP}, s{S_SOFF}, {4096 * i}"))
            ex[k].append(("vmem", f"buffer_load_dwordx4 {quad(RV0 + 4 * i)}, %{o['voff']}, s[{S_RS}:{S_RS + 3}], "
                                  f"s{S_TMP if i else S_SOFF} offen"))
        ex[8].append(("salu", f"s_add_u32 s{S_SOFF}, s{S_SOFF}, %{o['soff_step']}"))
    if S and not XOR_AT_END:
        ex[8].append(("valu", f"v_xor_b32 v{V_ST}, 0x{BUF_XOR:x}, v{V_ST}"))
    return ex


# ------------------------------------------------------------------------------------------------ the iteration
XOR_AT_END = _AB.get("KL_XOR_END", "1") == "1"


def body(st: Stream, extras: dict, tag: str, prio_head: int = 0, stores: bool = True):
    r"""One iteration.  `prio_head` > 0: the wave runs its first `prio_head` MFMA gaps at raised priority (the younger wave of a
    SIMD otherwise only runs when the older one stalls, so its loads would be issued late in the stage)."""
    st.drain()
    if "barrier" not in ABL:
        st.emit("s_barrier")
    if prio_head:
        st.emit("s_setprio 2")
    st.read_set(0, 0)
    st.read_set(1, 1)
    k = 0

    def put_extras(k):
        for kind, text in extras[k]:
            if (kind == "vmem" and ("vload" in ABL and "dwordx2" in text or "uload" in ABL and "dwordx4" in text)) or \
                    (kind == "valu" and "vtrans" in ABL and ("v_pk_add" in text or "v_sub_f32" in text or "v_add_f32" in text)) or \
                    (kind == "lds" and ("vstore" in ABL and "ds_write2st64" in text or "ustore" in ABL and "ds_write_b128" in text)):
                continue
            if kind == "lds":
                st.lds(text)
            else:  # "ldsx": self-contained LDS traffic that ends in its own lgkmcnt(0) (rare path)
                st.emit(text)
        if prio_head and k == prio_head - 1:
            st.emit("s_setprio 0")

    # carried groups of the previous stage (their reads completed before the barrier); none in a workgroup's first iteration
    for g, s in ((6, 2), (7, 3)):
        for j in range(4):
            st.emit(f"s_cmp_lg_u32 s{S_FIRST}, 0")
            st.emit(f"s_cbranch_scc1 Lsk{tag}{k}_%=")
            st.mfma(g, s, j)
            st.emit(f"Lsk{tag}{k}_%=:")
            put_extras(k)
            k += 1
    for f in range(6):
        if f + 2 < 8:
            st.read_set((f + 2) % 4, f + 2)
        st.need_set(f % 4)
        for j in range(4):
            st.mfma(f, f % 4, j)
            put_extras(k)
            k += 1
    st.emit(f"v_xor_b32 v{V_FA}, 0x{BUF_XOR:x}, v{V_FA}")
    st.emit(f"v_xor_b32 v{V_FB}, 0x{BUF_XOR:x}, v{V_FB}")
    if XOR_AT_END and stores:  # the store address toggles with the fragment addresses: one burst, one pipe switch
        st.emit(f"v_xor_b32 v{V_ST}, 0x{BUF_XOR:x}, v{V_ST}")
    st.emit(f"s_mov_b32 s{S_FIRST}, 0")


def flat(items):
    return [t for _, t in items]


ABL = set(_AB.get("KL_ABLATE", "").split(","))  # timing ablations (WRONG results): vload, uload, vtrans, vstore, ustore, barrier, afrag
# Tunables (environment overrides are for tools/kloop_variant.py A/B builds; the committed .inc is the default)
PRIO_HEAD = {"V": int(_AB.get("KL_PRIO_V", "0")), "U": int(_AB.get("KL_PRIO_U", "0"))}  # U role = waves 4..7 = the younger wave of every SIMD


def gen_role(role: str, affine: bool = False) -> list[str]:
    global AFFINE
    AFFINE = affine
    OOL.clear()
    V = role == "V"
    o = V_OPS if V else U_OPS
    extras = v_extras if V else u_extras
    st = Stream()
    e = st.emit
    P0 = TB  # the prologue's first stage is loaded into the (still unused) fragment registers, the second into RV:
    #          both global-memory latencies overlap

    def u_loads(rv0):
        for i in range(8):
            if i:
                e(f"s_add_u32 s{S_TMP}, s{S_SOFF}, {4096 * i}")
            e(f"buffer_load_dwordx4 {quad(rv0 + 4 * i)}, %{o['voff']}, s[{S_RS}:{S_RS + 3}], s{S_TMP if i else S_SOFF} offen")
        e(f"s_add_u32 s{S_SOFF}, s{S_SOFF}, %{o['soff_step']}")

    def v_loads(rv0, tag):
        for kind, t in v_load_events(tag):
            e(t)
        for i in range(16):
            e(v_load(i, rv0)[1])
        for t in flat(v_load_done()):
            e(t)

    # ---- initial state
    e(f"v_mov_b32 v{V_FA}, %{o['fragA']}")
    e(f"v_mov_b32 v{V_FB}, %{o['fragB']}")
    if V or not UDMA:
        e(f"v_mov_b32 v{V_ST}, %{o['st']}")
    for w in range(4):
        e(f"s_mov_b32 s{S_RS + w}, %{(o['rs0'] if V else o['rw']) + w}")
    e(f"s_mov_b32 s{S_SOFF}, %{o['soff0']}")
    e(f"s_mov_b32 s{S_FIRST}, 1")
    e(f"s_sub_u32 s{S_CNT}, %{o['kt_end']}, %{o['kt_begin']}")   # n stages
    if V:
        e(f"v_mov_b32 v{V_OOB}, 0x80000000")
        e(f"s_mov_b32 s{S_KT}, %{o['kt_begin']}")
        if affine:  # raw buffer over [scale | shift] (bounds unchecked: every lane reads channels of its own image), first pairs
            e(f"s_mov_b32 s{S_AF}, %{o['af_lo']}")
            e(f"s_and_b32 s{S_AF + 1}, %{o['af_hi']}, 0xffff")
            e(f"s_mov_b32 s{S_AF + 2}, 0x7fffffff")
            e(f"s_mov_b32 s{S_AF + 3}, 0x00020000")
            e(f"s_mov_b32 s{S_AFT}, %{o['af_delta']}")
            for kind, t in v_affine_loads("i", back=0):
                e(t)
        for q in range(4):
            e(f"ds_read_b128 {quad(VO0 + 4 * q)}, %{o['ldsA']} offset:{16 * q}")
        e("s_waitcnt lgkmcnt(0)")
        if affine:  # lane masks of the valid (non-padding) patch positions
            for i in range(16):
                e(f"v_cmp_le_i32 s[{S_MASK + 2 * i}:{S_MASK + 2 * i + 1}], 0, v{VO0 + i}")
    # ---- prologue: L(kt0) -> P0, L(kt0 + 1) -> RV (if n >= 2), S(kt0) from P0
    if V:
        v_loads(P0, "p0")
    elif UDMA:
        e(f"s_mov_b32 s{S_UST}, %{o['st']}")
        for i in range(8):
            for kind, t in u_dma(i, o):
                e(t)
        e(f"s_add_u32 s{S_SOFF}, s{S_SOFF}, %{o['soff_step']}")
        e(f"s_xor_b32 s{S_UST}, s{S_UST}, 0x{BUF_XOR:x}")
    else:
        u_loads(P0)
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e(f"s_cbranch_scc1 L{role}one_%=")
    if V:
        v_loads(RV0, "p1")
    elif not UDMA:
        u_loads(RV0)
    e(f"s_waitcnt vmcnt({16 if V else (0 if UDMA else 8)})")
    e(f"s_branch L{role}st0_%=")
    e(f"L{role}one_%=:")
    e("s_waitcnt vmcnt(0)")
    e(f"L{role}st0_%=:")
    if V:
        for c in range(4):
            for kind, t in v_affine_col(c, "p", P0):
                e(t)
            for t in v_pass1_col(c, P0):
                e(t)
        # (s_soff: one or two stages were requested; the second one's scale / shift sit 32 bytes behind it -- or nowhere
        # when the slice has a single stage: the load is then harmless and unused)
        for kind, t in v_affine_loads("q", back=32):
            e(t)
        for xi in range(4):
            valu, wa, wb = v_pass2_row(xi, P0)
            for kind, t in valu + [wa, wb]:
                e(t)
    elif not UDMA:
        for i in range(8):
            e(f"ds_write_b128 v{V_ST}, {quad(P0 + 4 * i)} offset:{4096 * i}")
    e(f"v_xor_b32 v{V_ST}, 0x{BUF_XOR:x}, v{V_ST}")
    # ---- n = 1: last; n = 2: pen, last; n >= 3: n - 2 steady iterations first
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e(f"s_cbranch_scc1 L{role}last_%=")
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
    e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    e(f"s_cbranch_scc1 L{role}pen_%=")
    e(".p2align 6")
    for _ in range(int(_AB.get("KL_PAD", "0"))):  # A/B: byte phase of the loop body inside its 64-byte line
        e("s_nop 0")
    e(f"L{role}steady_%=:")
    body(st, extras(True, True, "s"), "s", PRIO_HEAD[role])
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e(f"s_cbranch_scc1 L{role}steady_%=")
    # ---- second-to-last iteration: stores of the last stage, no loads
    e(f"L{role}pen_%=:")
    body(st, extras(True, False, "q"), "q", PRIO_HEAD[role])
    # ---- last iteration
    e(f"L{role}last_%=:")
    body(st, extras(False, False, "r"), "r", 0, stores=False)
    # ---- the last stage's groups 6, 7; every wave's fragment reads are complete behind this barrier, so the epilogue may
    #      reuse the stage buffers
    st.drain()
    e("s_barrier")
    for g, s in ((6, 2), (7, 3)):
        for j in range(4):
            st.mfma(g, s, j)
    e("s_nop 15")
    e("s_nop 7")
    if OOL:
        e(f"s_branch L{role}end_%=")
        for blk in OOL:
            for t in blk:
                e(t)
        e(f"L{role}end_%=:")
    return st.lines


def as_macro(name: str, lines: list[str]) -> str:
    out = [f"#define {name} \\"]
    for ln in lines:
        out.append(f'  "{ln}\\n" \\')
    out.append('  ""')
    return "\n".join(out) + "\n"


def clobbers() -> str:
    tail = ['"vcc"', '"scc"', '"memory"']
    plain = (['"m0"'] if UDMA else []) + [f'"v{i}"' for i in range(TB, 256)] + [f'"s{i}"' for i in range(S_FIRST, S_TMP + 1)] + tail
    aff = [f'"v{i}"' for i in range(SP, 256)] + [f'"s{i}"' for i in range(S_MASK, S_TMP + 1)] + tail
    return ("#define WINO_KLOOP_CLOBBERS " + ", ".join(plain) + "\n" + "#define WINO_KLOOP_VA_CLOBBERS " + ", ".join(aff) + "\n")


def generate() -> str:
    src = ("// generated by gen_wino_kloop.py -- do not edit.  Operand numbering: see V_OPS / U_OPS in the generator and the\n"
           "// asm statements in conv.hip.\n")
    src += as_macro("WINO_KLOOP_V_ASM", gen_role("V"))
    src += as_macro("WINO_KLOOP_VA_ASM", gen_role("V", affine=True))
    src += as_macro("WINO_KLOOP_U_ASM", gen_role("U"))
    src += clobbers()
    if UDMA:
        src += "#define WINO_KLOOP_UDMA 1\n"
    return src


def main(default_name: str, gen) -> int:
    import sys

    path = os.path.join(HERE, default_name)
    if "--out" in sys.argv:
        path = sys.argv[sys.argv.index("--out") + 1]
    text = gen()
    if "--check" in sys.argv:
        same = os.path.exists(path) and open(path).read() == text
        print(path, "up to date" if same else "DIFFERS from the generator's output")
        return 0 if same else 1
    with open(path, "w") as f:
        f.write(text)
    print(path, len(text.splitlines()), "lines")
    return 0


if __name__ == "__main__":
    raise SystemExit(main("wino_kloop.inc", generate))
