r"""Generates ``igemm_kloop.inc``: the K loop of ``conv_igemm_kernel<32>`` for its simplest and most frequent case -- ONE
tap (1 x 1 convolutions, ``nn.Linear`` on tokens, ``Conv1d(k=1)``), one or two sources, channel counts that are multiples of
the 32-channel K tile -- as a hand-scheduled gfx950 instruction stream (the Winograd kernel's recipe, gen_wino_kloop.py).

Why: ablations of the C++ loop on the 16384 x 768 -> 3072 token GEMM (round 3): shipped 670 us; without the LDS stores and the
barrier 555 us (+20 %); without the global loads 635 us.  The stall structure [loads][64 MFMAs][wait, stores][barrier][fragment
latency] costs a sixth of the kernel even with two independent workgroups per CU.  The stream:

    iteration t:   s_barrier                         S(t) visible; every R(t-1) done
                   R(t)[0], R(t)[1]                  fragments of k-groups 0, 1 (4 ds_read_b128 each: a0, a1, b0, b1)
                   M(t-1)[3]                         16 MFMAs whose fragments were read BEFORE the barrier
                   M(t)[0..2]                        with R(t)[2], R(t)[3] issued one group (16 MFMAs) ahead
                   beside the first 9 MFMAs:         S(t+1): s_waitcnt for L(t+1), 8 ds_write_b128 into the other stage;
                                                     L(t+2): 8 buffer_load_dwordx4 (4 filter rows, 4 pixel rows per thread)

Registers: v152..v254 (4 fragment sets of 16, 8 staging quads, 4 pixel-row offsets, 3 LDS addresses) and s84..s97, declared as clobbers; the four
accumulators, the 8 (+ 4 for a second source) per-lane row offsets, the buffer descriptors' words and a few scalars are asm
operands.  The address arithmetic in front of the loop and the fused epilogue stay C++ (conv.hip).

    python azula_amd/csrc/gen_igemm_kloop.py        # rewrites igemm_kloop.inc next to this file
"""
from __future__ import annotations

import os

from gen_wino_kloop import _AB, Stream, as_macro, main, quad  # the lgkmcnt bookkeeping and the macro writer are shared

HERE = os.path.dirname(os.path.abspath(__file__))

TB = 152           # fragment sets: TB + 16 s + {0: a0, 4: a1, 8: b0, 12: b1}   (v152..v215)
RV = 216           # 8 staging quads: filter rows 0..3, pixel rows 0..3           (v216..v247)
VOA = 248          # the 4 pixel-row offsets in use (first source: copied from the operands; replaced at the source switch)
V_FA, V_FB, V_ST = 252, 253, 254
S_RW, S_RS = 84, 88
S_SOFFW, S_SOFFA, S_CNT, S_FIRST, S_KT, S_TMP = 92, 93, 94, 95, 96, 97
S_NEXT, S_DELTA, S_TAP, S_KX, S_KS = 98, 99, 80, 81, 82   # taps variant: next tap boundary (stage index), byte offset of the tap, tap, kx, ks
V_TMP, V_OOB = 255, 151   # (taps variant)
TAPS = False  # set by gen(): the multi-tap variant (3 x 3 ... convolutions, ONE source, no upsampling, zero padding)
LS = 36            # floats per LDS row (conv.hip: KT + 4)
ROW32 = 32 * LS * 4          # bytes between a thread's loader rows (r0 + 32 i)
B_OFF = 128 * LS * 4         # the pixel tile sits behind the 128 filter rows

# operands
OPS = dict(acc=0, fragA=4, fragB=5, st=6, dA=7, dB=8, dS=9, voffW=10, voffA=14, voffA1=18, rw=22, rs0=26, rs1=30, n=34, kt0=35,
           kt_switch=36, soffW0=37, soffA0=38)


def frag(s, which, j=None):
    base = TB + 16 * s + 4 * which
    return f"v[{base}:{base + 3}]" if j is None else f"v{base + j}"


class GStream(Stream):
    def read_set(self, s: int, kk: int):
        off = kk * 32  # 8 floats per k-group
        self.lds(f"ds_read_b128 {frag(s, 0)}, v{V_FA} offset:{off}")
        self.lds(f"ds_read_b128 {frag(s, 1)}, v{V_FA} offset:{off + 32 * LS * 4}")
        self.lds(f"ds_read_b128 {frag(s, 2)}, v{V_FB} offset:{off}")
        self.lds(f"ds_read_b128 {frag(s, 3)}, v{V_FB} offset:{off + 32 * LS * 4}")
        self.set_done_at[s] = self.lds_seq

    def group(self, s: int, extras: dict, k0: int, skip_tag: str | None = None):
        r"""16 MFMAs of one k-group: 4 k-steps x (acc00, acc01, acc10, acc11)."""
        k = k0
        for ss in range(4):
            for a_i, b_i, acc in ((0, 2, 0), (0, 3, 1), (1, 2, 2), (1, 3, 3)):
                if skip_tag is not None:
                    self.emit(f"s_cmp_lg_u32 s{S_FIRST}, 0")
                    self.emit(f"s_cbranch_scc1 LGsk{skip_tag}{k}_%=")
                self.emit(f"v_mfma_f32_32x32x2_f32 %{acc}, {frag(s, a_i, ss)}, {frag(s, b_i, ss)}, %{acc}")
                if skip_tag is not None:
                    self.emit(f"LGsk{skip_tag}{k}_%=:")
                for kind, text in extras.get(k, []):
                    if kind == "lds":
                        self.lds(text)
                    else:
                        self.emit(text)
                k += 1


def load(i: int, rv0: int, second: bool = False):
    o = OPS
    if i < 4:
        return f"buffer_load_dwordx4 {quad(rv0 + 4 * i)}, %{o['voffW'] + i}, s[{S_RW}:{S_RW + 3}], s{S_SOFFW} offen"
    return f"buffer_load_dwordx4 {quad(rv0 + 4 * i)}, v{VOA + i - 4}, s[{S_RS}:{S_RS + 3}], s{S_SOFFA} offen"



def store(i: int, rv0: int):
    off = (i % 4) * ROW32 + (B_OFF if i >= 4 else 0)
    return f"ds_write_b128 v{V_ST}, {quad(rv0 + 4 * i)} offset:{off}"


def switch_event(tag: str):
    r"""In front of the pixel-row loads of stage s_kt: the second source starts (descriptor, row offsets, channel offset 0)."""
    o = OPS
    L = [("salu", f"s_cmp_eq_u32 s{S_KT}, %{o['kt_switch']}"), ("salu", f"s_cbranch_scc0 LGns{tag}_%=")]
    for w in range(4):
        L.append(("salu", f"s_mov_b32 s{S_RS + w}, %{o['rs1'] + w}"))
    L.append(("salu", f"s_mov_b32 s{S_SOFFA}, 0"))
    for i in range(4):
        L.append(("valu", f"v_mov_b32 v{VOA + i}, %{o['voffA1'] + i}"))
    L.append(("label", f"LGns{tag}_%=:"))
    return L


def tap_offsets():
    r"""The 4 pixel-row offsets of the current tap: base + tap offset where the tap is inside the image for that row (bit s_tap of
    the row's mask), the out-of-bounds offset (the load returns 0: zero padding) elsewhere."""
    o = OPS
    L = []
    for i in range(4):
        L.append(("valu", f"v_add_u32 v{VOA + i}, s{S_DELTA}, %{o['voffA'] + i}"))
        L.append(("valu", f"v_lshrrev_b32 v{V_TMP}, s{S_TAP}, %{o['voffA1'] + i}"))
        L.append(("valu", f"v_and_b32 v{V_TMP}, 1, v{V_TMP}"))
        L.append(("valu", f"v_cmp_ne_u32 vcc, 0, v{V_TMP}"))
        L.append(("valu", f"v_cndmask_b32 v{VOA + i}, v{V_OOB}, v{VOA + i}, vcc"))
    return L


def tap_event(tag: str):
    r"""Taps variant, in front of the loads of stage s_kt: the next tap starts -- its weights sit one (cout_s x cin_s) plane
    further (rs1 word 1 = that step minus what the channel walk added), the channel offset restarts, the pixel rows move by
    one column (cs4 bytes) or to the next row of the window."""
    o = OPS
    L = [("salu", f"s_cmp_eq_u32 s{S_KT}, s{S_NEXT}"), ("salu", f"s_cbranch_scc0 LGnt{tag}_%=")]
    L.append(("salu", f"s_add_u32 s{S_NEXT}, s{S_NEXT}, %{o['rs1']}"))
    L.append(("salu", f"s_add_u32 s{S_SOFFW}, s{S_SOFFW}, %{o['rs1'] + 1}"))
    L.append(("salu", f"s_mov_b32 s{S_SOFFA}, 0"))
    L.append(("salu", f"s_add_u32 s{S_TAP}, s{S_TAP}, 1"))
    L.append(("salu", f"s_add_u32 s{S_KX}, s{S_KX}, 1"))
    L.append(("salu", f"s_add_u32 s{S_DELTA}, s{S_DELTA}, %{o['rs1'] + 2}"))
    L.append(("salu", f"s_cmp_eq_u32 s{S_KX}, s{S_KS}"))
    L.append(("salu", f"s_cbranch_scc0 LGsr{tag}_%="))
    L.append(("salu", f"s_mov_b32 s{S_KX}, 0"))
    L.append(("salu", f"s_add_u32 s{S_DELTA}, s{S_DELTA}, %{o['rs1'] + 3}"))
    L.append(("label", f"LGsr{tag}_%=:"))
    L += tap_offsets()
    L.append(("label", f"LGnt{tag}_%=:"))
    return L


def load_done():
    return [("salu", f"s_add_u32 s{S_SOFFW}, s{S_SOFFW}, 128"), ("salu", f"s_add_u32 s{S_SOFFA}, s{S_SOFFA}, 128"),
            ("salu", f"s_add_u32 s{S_KT}, s{S_KT}, 1")]


XOR_AT_END = _AB.get("KG_XOR_END", "1") == "1"  # the store-address toggle beside the fragment-address toggles: one burst


def extras(S: bool, L: bool, tag: str):
    ex = {k: [] for k in range(64)}
    if S:
        ex[0].append(("wait", "s_waitcnt vmcnt(0)"))
        for i in range(8):
            ex[i].append(("lds", store(i, RV)))
        if not XOR_AT_END:
            ex[8].append(("valu", f"v_xor_b32 v{V_ST}, v{V_ST}, %{OPS['dS']}"))
    if L:
        for i in range(8):
            if i == 0 and TAPS:
                ex[1 + i] += tap_event(tag)
            if i == 4 and not TAPS:
                ex[1 + i] += switch_event(tag)
            ex[1 + i].append(("vmem", load(i, RV)))
        ex[9] += load_done()
    return ex


def body(st: GStream, ex: dict, tag: str, stores: bool = True):
    st.drain()
    st.emit("s_barrier")
    st.read_set(0, 0)
    st.read_set(1, 1)
    st.group(3, ex, 0, skip_tag=tag)       # carried: the previous stage's last k-group (none in the first iteration)
    st.read_set(2, 2)
    st.need_set(0)
    st.group(0, ex, 16)
    st.read_set(3, 3)
    st.need_set(1)
    st.group(1, ex, 32)
    st.need_set(2)
    st.group(2, ex, 48)
    st.emit(f"v_xor_b32 v{V_FA}, v{V_FA}, %{OPS['dA']}")
    st.emit(f"v_xor_b32 v{V_FB}, v{V_FB}, %{OPS['dB']}")
    if XOR_AT_END and stores:
        st.emit(f"v_xor_b32 v{V_ST}, v{V_ST}, %{OPS['dS']}")
    st.emit(f"s_mov_b32 s{S_FIRST}, 0")


def gen(taps: bool = False) -> list[str]:
    global TAPS
    TAPS = taps
    o = OPS
    st = GStream()
    e = st.emit
    P0 = TB  # the prologue's first stage lands in fragment sets 0, 1 (8 quads), the second in RV: both latencies overlap
    e(f"v_mov_b32 v{V_FA}, %{o['fragA']}")
    e(f"v_mov_b32 v{V_FB}, %{o['fragB']}")
    e(f"v_mov_b32 v{V_ST}, %{o['st']}")
    if not taps:
        for i in range(4):
            e(f"v_mov_b32 v{VOA + i}, %{o['voffA'] + i}")
    for w in range(4):
        e(f"s_mov_b32 s{S_RW + w}, %{o['rw'] + w}")
        e(f"s_mov_b32 s{S_RS + w}, %{o['rs0'] + w}")
    if taps:  # operand kt_switch = ks | tap0 << 8 | kx0 << 16 | ky0 << 24; rs1 words = (stages per tap, weight step, cs4, row step)
        e(f"v_mov_b32 v{V_OOB}, 0x80000000")
        e(f"s_and_b32 s{S_KS}, %{o['kt_switch']}, 0xff")
        e(f"s_lshr_b32 s{S_TAP}, %{o['kt_switch']}, 8")
        e(f"s_and_b32 s{S_TAP}, s{S_TAP}, 0xff")
        e(f"s_lshr_b32 s{S_KX}, %{o['kt_switch']}, 16")
        e(f"s_and_b32 s{S_KX}, s{S_KX}, 0xff")
        e(f"s_lshr_b32 s{S_TMP}, %{o['kt_switch']}, 24")             # ky0
        e(f"s_mul_i32 s{S_DELTA}, s{S_KS}, %{o['rs1'] + 2}")          # ks * cs4
        e(f"s_add_u32 s{S_DELTA}, s{S_DELTA}, %{o['rs1'] + 3}")      # + row step = one image row in bytes
        e(f"s_mul_i32 s{S_DELTA}, s{S_DELTA}, s{S_TMP}")              # ky0 rows
        e(f"s_mul_i32 s{S_TMP}, s{S_KX}, %{o['rs1'] + 2}")
        e(f"s_add_u32 s{S_DELTA}, s{S_DELTA}, s{S_TMP}")              # + kx0 columns
        e(f"s_add_u32 s{S_NEXT}, s{S_TAP}, 1")
        e(f"s_mul_i32 s{S_NEXT}, s{S_NEXT}, %{o['rs1']}")             # first stage of the next tap
        for kind, t in tap_offsets():
            e(t)
    e(f"s_mov_b32 s{S_SOFFW}, %{o['soffW0']}")
    e(f"s_mov_b32 s{S_SOFFA}, %{o['soffA0']}")
    e(f"s_mov_b32 s{S_KT}, %{o['kt0']}")
    e(f"s_mov_b32 s{S_FIRST}, 1")
    e(f"s_mov_b32 s{S_CNT}, %{o['n']}")

    def loads(rv0, tag):
        for i in range(8):
            if i == 0 and taps:
                for kind, t in tap_event(tag):
                    e(t)
            if i == 4 and not taps:
                for kind, t in switch_event(tag):
                    e(t)
            e(load(i, rv0))
        for kind, t in load_done():
            e(t)

    loads(P0, "p0")
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e("s_cbranch_scc1 LGone_%=")
    loads(RV, "p1")
    e("s_waitcnt vmcnt(8)")
    e("s_branch LGst0_%=")
    e("LGone_%=:")
    e("s_waitcnt vmcnt(0)")
    e("LGst0_%=:")
    for i in range(8):
        e(store(i, P0))
    e(f"v_xor_b32 v{V_ST}, v{V_ST}, %{o['dS']}")
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e("s_cbranch_scc1 LGlast_%=")
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
    e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 LGpen_%=")
    e(".p2align 6")
    e("LGsteady_%=:")
    body(st, extras(True, True, "s"), "s")
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 LGsteady_%=")
    e("LGpen_%=:")
    body(st, extras(True, False, "q"), "q")
    e("LGlast_%=:")
    body(st, extras(False, False, "r"), "r", stores=False)
    # the last stage's k-group 3; every wave's fragment reads are complete behind this barrier (the epilogue reuses the stages)
    st.drain()
    e("s_barrier")
    st.group(3, {}, 0)
    e("s_nop 15")
    e("s_nop 7")
    return st.lines


def clobbers() -> str:
    tail = ['"vcc"', '"scc"', '"memory"']
    regs = [f'"v{i}"' for i in range(TB, 256)] + [f'"s{i}"' for i in range(S_RW, S_TMP + 1)] + tail
    taps = [f'"v{i}"' for i in range(V_OOB, 256)] + [f'"s{i}"' for i in range(S_TAP, S_DELTA + 1)] + tail
    return "#define IGEMM_KLOOP_CLOBBERS " + ", ".join(regs) + "\n#define IGEMM_KLOOP_TAPS_CLOBBERS " + ", ".join(taps) + "\n"


def generate() -> str:
    src = "// generated by gen_igemm_kloop.py -- do not edit.  Operand numbering: OPS in the generator, the asm statement in conv.hip.\n"
    src += as_macro("IGEMM_KLOOP_ASM", gen())
    src += as_macro("IGEMM_KLOOP_TAPS_ASM", gen(taps=True))
    src += clobbers()
    return src


if __name__ == "__main__":
    raise SystemExit(main("igemm_kloop.inc", generate))
