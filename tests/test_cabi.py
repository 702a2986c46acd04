r"""The C-ABI shared library loads and exports every symbol ``include/azula_amd.h`` declares
(no compute calls: this runs without a GPU)."""

import ctypes
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "azula_amd.h")


def declared_symbols() -> list[str]:
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from azula_amd.csrc import build

    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    handle = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(handle, s)]
    assert not missing, f"declared in azula_amd.h but not exported: {missing}"


def test_python_prototypes_match_header(built_lib):
    from azula_amd import _lib

    syms = set(declared_symbols())
    assert set(_lib.PROTOTYPES) <= syms
    assert syms - set(_lib.PROTOTYPES) <= {"az_error_string"}
    assert _lib.lib().az_version() == 1
    assert b"NULL" in _lib.lib().az_error_string(-1)


def test_struct_layouts_match_c():
    from azula_amd import _lib

    names = ["AzStepCoef", "AzTransitionArgs", "AzMultistepArgs", "AzLinearGroup", "AzNormFinalizeArgs", "AzConvArgs",
             "AzAttnArgs"]
    prog = '#include <stdio.h>\n#include "azula_amd.h"\nint main(void){' + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names
    ) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    for n, sz in zip(names, sizes):
        assert ctypes.sizeof(getattr(_lib, n)) == sz, n


def test_missing_library_is_loud(monkeypatch):
    from azula_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libazula_amd.so")
    with pytest.raises(_lib.AzulaAmdError, match="no CPU/eager fallback"):
        _lib.lib()


def test_argument_errors_are_returned_not_raised(built_lib):
    """Every entry point validates its arguments BEFORE touching the device and reports through its int status
    (never throws, never exits) -- checkable without a GPU.  Negative codes are AZ_E_*; az_error_string names them."""
    from azula_amd import _lib

    lib = _lib.lib()
    assert lib.az_scale_f32(None, None, None, 16, None) == -1  # AZ_E_NULL
    assert lib.az_axpby_f32(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0, 16, 0, None) == -2  # rows == 0: AZ_E_SHAPE
    att = _lib.AzAttnArgs(q=0x1000, k=0x1000, v=0x1000, out=0x1000, batch=1, heads=1, tokens=8, head_dim=48)
    assert lib.az_attention_f32(ctypes.byref(att), None) == -4  # head_dim 48: AZ_E_UNSUPPORTED
    conv = _lib.AzConvArgs(src0=0x1008, weight=0x1000, dst=0x1000, c0s=8, cout_s=8, batch=1, hin=4, win=4, hout=4, wout=4,
                           ksize=3, stride=1, pad=1, splitk=1, h0=4, w0=4)
    assert lib.az_conv2d_f32(ctypes.byref(conv), None) == -3  # src0 not 16-byte aligned: AZ_E_ALIGN
    assert lib.az_conv2d_x3_f32(ctypes.byref(conv), None) == -3  # the bf16x3 entry shares the validation
    assert lib.az_pack_conv_weight_x3_f32(None, 0x1000, 8, 8, 3, 8, 8, 8, 8, None) == -1
    assert lib.az_pack_conv_weight_x3_f32(0x1000, 0x1000, 8, 8, 3, 6, 8, 8, 8, None) == -2  # cout_s < cout / not % 4
    conv.src0, conv.stride = 0x1000, 2
    assert lib.az_conv2d_winograd_f32(ctypes.byref(conv), None) == -4  # Winograd is stride 1 only
    ms = _lib.AzMultistepArgs(x_s=0x1000, pred=0x1000, x_t=0x1000, mean=0x1000, coef=0x1000, count=16, n_hist=9)
    assert lib.az_multistep_f32(ctypes.byref(ms), None) == -2
    # round 6: typed (half-precision) tensors are a property of the bf16 / f16 entries only; the typed passes say what they do not take
    conv.stride, conv.src0 = 1, 0x1000
    conv.dst_dtype = 1
    assert lib.az_conv2d_f32(ctypes.byref(conv), None) == -4 and lib.az_conv2d_x3_f32(ctypes.byref(conv), None) == -4
    assert lib.az_conv2d_winograd_x3_f32(ctypes.byref(conv), None) == -4
    conv.dst_dtype, conv.src_dtype, conv.c0s = 0, 1, 12
    assert lib.az_conv2d_bf16_f32(ctypes.byref(conv), None) == -2  # 2-byte sources: channel strides in multiples of 8
    att = _lib.AzAttnArgs(q=0x1000, k=0x1000, v=0x1000, out=0x1000, batch=1, heads=1, tokens=8, head_dim=64, io_dtype=1)
    assert lib.az_attention_f32(ctypes.byref(att), None) == -4 and lib.az_attention_x3_f32(ctypes.byref(att), None) == -4
    att.io_dtype, att.norm_dim = 0, 65
    assert lib.az_attention_f32(ctypes.byref(att), None) == -2  # norm_dim <= head_dim
    assert lib.az_rownorm_mod_h16(0x1000, 0x1000, None, None, None, 0, 4, 4, 20, 20, 1, 1e-5, 1, None) == -4  # C % 8
    assert lib.az_rownorm_mod_h16(0x1000, 0x1000, None, None, None, 0, 4, 4, 64, 64, 1, 1e-5, 3, None) == -2  # dtype 1 | 2
    assert lib.az_groupnorm_stats_h16(0x1000, 0x1000, None, 0, 1, 16, 24, 24, 8, 1, 1, None) == -4  # groups of 3 channels: no typed form
    assert lib.az_affine_act_h16(0x1000, 0x1000, None, 0, 0x1000, 0x1000, 1, 4, 4, 12, 0, 0, 1, None) == -2  # cs % 8
    assert lib.az_token_fill_h16(0x1000, 8, 0, 4, 0x1000, 64, 0x1000, 1, 60, 1, None) == -2
    # round 6: the f16x2 entries share the x3 entries' validation and add their own (the weight scale is a power of two, the absmax
    # slots are 16-byte aligned and in_absmax1 comes only with in_absmax0)
    conv = _lib.AzConvArgs(src0=0x1000, weight=0x1000, dst=0x1000, c0s=8, cout_s=8, batch=1, hin=4, win=4, hout=4, wout=4,
                           ksize=3, stride=1, pad=1, splitk=1, h0=4, w0=4, w_scale=3.0)
    assert lib.az_conv2d_f16x2_f32(ctypes.byref(conv), None) == -2 and lib.az_conv2d_winograd_f16x2_f32(ctypes.byref(conv), None) == -2
    conv.w_scale, conv.in_absmax0 = 1024.0, 0x1004
    assert lib.az_conv2d_f16x2_f32(ctypes.byref(conv), None) == -3 and lib.az_conv2d_winograd_f16x2_f32(ctypes.byref(conv), None) == -3
    conv.in_absmax0, conv.in_absmax1 = None, 0x1000
    assert lib.az_conv2d_f16x2_f32(ctypes.byref(conv), None) == -3
    conv.in_absmax1, conv.dst_dtype = None, 1
    assert lib.az_conv2d_f16x2_f32(ctypes.byref(conv), None) == -4  # typed tensors: the bf16 / f16 entries only
    assert lib.az_pack_conv_weight_f16x2_f32(0x1000, 0x1000, 8, 8, 3, 8, 8, 8, 8, 3.0, None) == -2
    assert lib.az_winograd_pack_filter_f16x2_f32(0x1000, 0x1000, 8, 8, 8, 1, 1, 1, 0.0, None) == -2
    assert lib.az_absmax_f32(None, 0x1000, 16, None) == -1 and lib.az_absmax_f32(0x1000, 0x1000, 0, None) == -2 and lib.az_absmax_f32(0x1000, 0x1004, 16, None) == -3
    assert lib.az_absmax_from_moments_f32(0x1000, None, 4, None) == -1 and lib.az_absmax_from_moments_f32(0x1000, 0x1000, 0, None) == -2
    att = _lib.AzAttnArgs(q=0x1000, k=0x1000, v=0x1000, out=0x1000, batch=1, heads=1, tokens=8, head_dim=48)
    assert lib.az_attention_f16x2_f32(ctypes.byref(att), None) == -4
    # the weight scale the packings want: amax (x 2.25 for the Winograd transform) lands in [2^13, 2^14); degenerate maxima -> 1
    ws = lib.az_f16x2_weight_scale
    assert ws(1.0, 0) == 8192.0 and ws(0.9, 0) == 16384.0 and ws(1.0, 1) == 4096.0 and ws(0.0, 0) == 1.0 and ws(float("inf"), 0) == 1.0
    assert 8192.0 <= 0.013 * ws(0.013, 0) < 16384.0 and 8192.0 <= 2.25 * 300.0 * ws(300.0, 1) < 16384.0
    for code, word in ((-1, b"NULL"), (-2, b"shape"), (-3, b"align"), (-4, b"unsupported")):
        assert word.lower() in lib.az_error_string(code).lower()
    with pytest.raises(_lib.AzulaAmdError, match="az_scale_f32"):
        _lib.call("az_scale_f32", None, None, None, 16, None)


def test_committed_kloop_streams_are_the_generators_output(monkeypatch):
    r"""wino_kloop.inc / igemm_kloop.inc are committed sources that a build never rewrites: they must be exactly what their
    generators emit, and a user's KL_* / KG_* environment must not change that (only AZ_KLOOP_AB=1 A/B builds honour them)."""
    import subprocess
    import sys

    csrc = os.path.join(ROOT, "azula_amd", "csrc")
    env = dict(os.environ, KL_ABLATE="vload,uload", KL_SCALAR_ADD="1", KG_XOR_END="0")
    env.pop("AZ_KLOOP_AB", None)
    for gen in ("gen_wino_kloop.py", "gen_igemm_kloop.py"):
        res = subprocess.run([sys.executable, os.path.join(csrc, gen), "--check"], env=env, capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr
    tracked = subprocess.run(["git", "ls-files", "azula_amd/csrc/_obj", "azula_amd/csrc/_ab"], cwd=ROOT, capture_output=True, text=True)
    assert tracked.returncode != 0 or tracked.stdout.strip() == "", "build by-products are tracked: " + tracked.stdout


def test_ablation_variants_apply_to_the_committed_sources():
    r"""tools/ablate.py builds timing variants from textual substitutions on a COPY of csrc: every pattern of every variant must occur
    exactly once in the committed sources (two variants had gone stale behind kernel edits, so profiles cited in
    DESIGN could not be reproduced).  No compilation here -- only that the patches still apply."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("az_ablate", os.path.join(ROOT, "tools", "ablate.py"))
    ab = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ab)
    csrc = os.path.join(ROOT, "azula_amd", "csrc")
    assert len(ab.VARIANTS) >= 40
    for name, patches in ab.VARIANTS.items():
        texts: dict = {}
        for f, old, new in patches:
            t = texts[f] if f in texts else open(os.path.join(csrc, f)).read()
            assert t.count(old) == 1, f"variant {name}: pattern occurs {t.count(old)} times in {f}: {old[:70]!r}"
            texts[f] = t.replace(old, new)


def test_no_ungated_environment_switch_in_the_launchers():
    r"""kernel selection reads the environment only through az_ab_env (honoured under AZ_DEBUG_AB), and the
    launchers keep no unsynchronised static flag (the LDS attribute is set per (kernel, device) through an atomic mask)."""
    import re

    csrc = os.path.join(ROOT, "azula_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h", ".inc")):
            continue
        text = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", text)
        n_env = len(re.findall(r"\bgetenv\s*\(", code))
        assert n_env == (2 if f == "common.h" else 0), (f, n_env)  # (common.h: az_ab_env itself)
        assert not re.search(r"static\s+bool\s+\w+\s*=\s*false", code), f
        assert "hipFuncSetAttribute" not in code or f == "common.h", f


def test_bf16x3_tile_plan_on_the_host(built_lib, monkeypatch):
    """az_conv2d_x3_suggest_splitk is host arithmetic (no device): it pins the 256 x 256-tile plan of the bf16x3 token GEMMs --
    one workgroup per CU, so whole rounds of the 256 CUs, never a split of a grid that already fills 0.75 of a round, a split
    in two for a deep K under one round -- and falls back to az_conv2d_suggest_splitk wherever the 128 x 128 kernel runs."""
    from azula_amd import _lib

    lib = _lib.lib()
    monkeypatch.delenv("AZ_X3_BIG", raising=False)

    def gemm(tokens, cin, cout, **kw):
        d = dict(src0=0x1000, weight=0x1000, dst=0x1000, c0s=cin, cout_s=cout, batch=1, hin=tokens, win=1, hout=tokens, wout=1,
                 ksize=1, stride=1, pad=0, splitk=1, h0=tokens, w0=1)
        d.update(kw)
        return _lib.AzConvArgs(**d)

    sk = lambda a: lib.az_conv2d_x3_suggest_splitk(ctypes.byref(a))  # noqa: E731
    assert sk(gemm(16384, 768, 3072)) == 1   # 768 big tiles = 3 whole rounds
    assert sk(gemm(16384, 3072, 768)) == 1   # 192 big tiles, deep K: unsplit (the 128 x 128 kernel would split in two)
    assert lib.az_conv2d_suggest_splitk(16384, 768, 3072, 1) == 2
    assert sk(gemm(9216, 2048, 768)) == 2    # 108 big tiles: the deep K loop in two halves -> 216 workgroups
    assert sk(gemm(9216, 768, 768)) == lib.az_conv2d_suggest_splitk(9216, 768, 768, 1)      # shallow K, under a round: 128 x 128 tiles
    assert sk(gemm(256, 1024, 1024)) == lib.az_conv2d_suggest_splitk(256, 1024, 1024, 1)    # a small map: 128 x 128 tiles, their split-K
    assert sk(gemm(58 * 256, 1536, 768)) == 1  # 4 x 192-cout tiles x 58 = 232 workgroups (what the launch picks), not 174 256-cout tiles split in two
    conv3 = gemm(16384, 768, 3072, ksize=3, pad=1)
    assert sk(conv3) == lib.az_conv2d_suggest_splitk(16384, 3072, 768, 3)                    # taps: not a big-tile launch
    monkeypatch.setenv("AZ_X3_BIG", "0")
    monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
    assert sk(gemm(16384, 3072, 768)) == 2
