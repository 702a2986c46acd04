r"""Attention kernel + ViT/DiT backbone parity on the GPU (oracle / reference golden vectors)."""

import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import max_err
from oracle import nets, sampling, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("B,H,T,D", [(2, 4, 16, 16), (1, 2, 64, 32), (2, 3, 256, 64), (1, 2, 100, 64), (1, 1, 1024, 64), (1, 2, 40, 128), (2, 2, 72, 80)])
@pytest.mark.parametrize("order,rms", [("nHC", True), ("nHC", False), ("H3C", False), ("3HC", False)])
@pytest.mark.parametrize("x3", [True, False, "f16x2"])
def test_attention_kernel(monkeypatch, B, H, T, D, order, rms, x3):
    """az_attention_x3_f32 (the contractions as 3 x bf16 pieces / 6 partial products on the bf16 MFMA), az_attention_f16x2_f32
    (2 x f16 pieces / 3 partial products on the f16 MFMA) and az_attention_f32 (fp32 MFMA), all against torch's SDPA at the same bound."""
    from azula_amd import engine
    from azula_amd.engine import Act, Builder

    monkeypatch.setattr(engine, "ATTN_X3", bool(x3))
    monkeypatch.setattr(engine, "FP32_MFMA", "f16x2" if x3 == "f16x2" else "bf16x3")
    xname = "az_attention_f16x2_f32" if x3 == "f16x2" else "az_attention_x3_f32"

    g = torch.Generator().manual_seed(B * T + D)
    q, k, v = (torch.randn(B, H, T, D, generator=g) for _ in range(3))
    if order in ("nHC", "3HC"):
        qkv = torch.stack((q, k, v), dim=0).permute(1, 3, 0, 2, 4).reshape(B, T, 3 * H * D)  # (n H C)
    else:
        qkv = torch.stack((q, k, v), dim=2).permute(0, 3, 1, 2, 4).reshape(B, T, 3 * H * D)  # (H 3 C)
    qn, kn = (F.rms_norm(q, (D,), eps=1e-5), F.rms_norm(k, (D,), eps=1e-5)) if rms else (q, k)
    ref = F.scaled_dot_product_attention(qn, kn, v)  # default scale 1/sqrt(D)
    ref = ref.transpose(1, 2).reshape(B, T, H * D)
    bld = Builder(torch.device("cuda"))
    act = Act(qkv.cuda().contiguous().reshape(-1), B, T, 1, 3 * H * D, 3 * H * D, True)
    act.bounded = True  # (a projection of normalised tokens: what the f16x2 form is chosen for)
    out = bld.attention(act, H, order, rms, 1.0 / math.sqrt(D))
    assert [n for _, _, n in bld.tape.ops] == [xname if x3 and D <= 80 else "az_attention_f32"]
    if x3 and D > 80:  # the engine keeps head_dim 128 on the fp32 kernel (faster there); the entry point itself takes it
        from azula_amd import _lib

        _, args, _ = bld.tape.ops[0]
        bld.tape.ops[0] = (getattr(_lib.lib(), xname), args, xname)
    bld.tape.run()
    got = out.buf.reshape(B, T, H * D)
    assert max_err(got, ref) < 2e-5, max_err(got, ref)


@pytest.mark.parametrize("B,H,T,D", [(2, 4, 48, 32), (2, 3, 288, 64), (1, 2, 100, 128), (3, 12, 64, 64)])
@pytest.mark.parametrize("rms,gains,rope", [(True, False, False), (True, True, True), (False, False, True)])
@pytest.mark.parametrize("half", [None, torch.bfloat16])
def test_qk_preparation_in_the_projection_epilogue(B, H, T, D, rms, gains, rope, half):
    """AzConvArgs.act = 5: the fused q | k | v projection's epilogue applies the q / k RMS norm, the learned gains and RoPE
    once per layer (azula/nn/attention.py:92-95); the attention kernel then takes q and k as they are.  Checked against
    torch (linear -> rms_norm -> gain -> rotate -> SDPA) and against the in-kernel path (AZ_QK_PREP off)."""
    from azula_amd import engine
    from azula_amd.engine import Act, Builder

    Cin, HC = 64, H * D
    g = torch.Generator().manual_seed(B * T + D + H)
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(3 * HC, Cin, generator=g) / math.sqrt(Cin)
    bias = torch.randn(3 * HC, generator=g) * 0.1
    qw, kw = (1 + 0.2 * torch.randn(D, generator=g) for _ in range(2))
    theta = torch.randn(T, H * D // 2, generator=g) * 2
    qkv = F.linear(x, w, bias)
    q, k, v = (t.reshape(B, T, H, D).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    if rms:
        q, k = F.rms_norm(q, (D,), eps=1e-6), F.rms_norm(k, (D,), eps=1e-6)
    if gains:
        q, k = q * qw, k * kw
    if rope:
        cs, sn = (f(theta).reshape(T, H, D // 2).transpose(0, 1) for f in (torch.cos, torch.sin))

        def rot(t):
            re, im = t[..., 0::2], t[..., 1::2]
            return torch.stack((re * cs - im * sn, re * sn + im * cs), dim=-1).flatten(-2)

        q, k = rot(q), rot(k)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, HC)
    outs = {}
    for mode in (True, False):
        engine.QK_PREP = mode
        try:
            bld = Builder(torch.device("cuda"))
            bld.half = half
            xa = Act(x.cuda().reshape(-1), B, T, 1, Cin, Cin, True)
            tabs = (bld.const(torch.cos(theta)), bld.const(torch.sin(theta))) if rope else None
            gw = (bld.const(qw), bld.const(kw)) if gains else None
            prep = dict(heads=H, head_dim=D, rmsnorm=rms, eps=1e-6, rope=tabs, weight=gw)
            y = bld.conv(xa, bld.pack_conv(w.cuda(), bias.cuda()), 3 * HC, qk_prep=prep)
            assert y.qk_prepared == mode and bld.tape.keep and (bld.tape.ops[-1][1][0]._obj.act == (5 if mode else 0))
            out = bld.attention(y, H, "3HC", rms, 1.0 / math.sqrt(D), eps=1e-6, rope=tabs, qk_weight=gw)
            bld.finish()
            bld.tape.run()
            outs[mode] = out.buf.reshape(B, T, HC).clone()
        finally:
            engine.QK_PREP = True
    tol = 3e-5 if half is None else 3e-2
    e_ref, e_ab = max_err(outs[True], ref), max_err(outs[True], outs[False])
    assert e_ref < tol and e_ab < tol, (e_ref, e_ab)


def test_attention_spiked_scores():
    """Online-softmax rescale path: one key dominates from a late tile (guide rule 26)."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(0)
    B, H, T, D = 1, 1, 256, 64
    q, k, v = (torch.randn(B, H, T, D, generator=g) for _ in range(3))
    k[0, 0, 200] = 6.0 * q[0, 0, 3]  # spike in the 4th key tile for query 3
    qkv = torch.stack((q, k, v), dim=0).permute(1, 3, 0, 2, 4).reshape(B, T, 3 * D)
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).float().transpose(1, 2).reshape(B, T, D)
    bld = Builder(torch.device("cuda"))
    out = bld.attention(Act(qkv.cuda().contiguous().reshape(-1), B, T, 1, 3 * D, 3 * D, True), 1, "nHC", False, 1 / 8.0)
    bld.tape.run()
    assert max_err(out.buf.reshape(B, T, D), ref) < 2e-5


def test_patchify_roundtrip():
    from azula_amd import _lib

    x = torch.randn(2, 3, 8, 12, device="cuda")
    p, cs = 2, 12
    tok = torch.empty(2, 4 * 6, cs, device="cuda")
    _lib.call("az_patchify_f32", tok.data_ptr(), x.data_ptr(), None, 2, 3, 8, 12, p, cs, _lib.stream_ptr())
    ref = x.reshape(2, 3, 4, p, 6, p).permute(0, 2, 4, 1, 3, 5).reshape(2, 24, 12)
    assert torch.equal(tok, ref)
    back = torch.empty_like(x)
    _lib.call("az_unpatchify_f32", back.data_ptr(), tok.data_ptr(), 2, 3, 8, 12, p, cs, _lib.stream_ptr())
    assert torch.equal(back, x)


def build_vit(cfg):
    from azula_amd.nn import ViT

    return ViT(
        cfg["in_channels"], cfg["out_channels"], hid_channels=cfg["hid_channels"], hid_blocks=cfg["hid_blocks"],
        attention_heads=cfg["attention_heads"], patch_size=cfg["patch_size"], mod_features=cfg["mod_features"],
    )


def test_vit_forward_matches_reference(golden):
    g = golden("g5_vit")
    cfg = g.meta["cfg"]
    net = build_vit(cfg)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in g.meta["shapes"].items()}
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    net = net.cuda().eval()
    x = g["x"].cuda()
    for tag in ("modB", "mod1"):
        y = net(x, g[tag].cuda())
        err, sc = max_err(y, g["y_" + tag]), g["y_" + tag].abs().max().item()
        print("vit", tag, "max|d|", err, "scale", sc)
        assert err < 1e-4 * max(1.0, sc)


@pytest.mark.parametrize("mode,entry", [("bf16x3", "az_conv2d_x3_f32"), ("f16x2", "az_conv2d_f16x2_f32")])
def test_vit_ddim50_through_bf16x3(golden, monkeypatch, mode, entry):
    """DDIM-50 of the golden ViT with every token GEMM on the bf16 MFMA as 3 x bf16 pieces / 6 partial products
    (AZ_FP32_MFMA=bf16x3) or on the f16 MFMA as 2 x f16 pieces / 3 partial products (f16x2): same tolerance as the native fp32 run
    (hid_channels = 64 >= 32, so the GEMMs qualify)."""
    from azula_amd import engine
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler

    monkeypatch.setattr(engine, "FP32_MFMA", mode)
    g = golden("g6_vit_loop")
    cfg = g.meta["cfg"]
    w = TimeModulated(build_vit(cfg), cfg["mod_features"], name="vit")
    w.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    den = KarrasDenoiser(w, VPSchedule()).cuda().eval()
    smp = DDIMSampler(den, steps=50, silent=True)
    x0 = smp(g["x1"].cuda())
    loop = next(iter(smp._fused_cache.values()))
    assert any(name_ == entry for _, _, name_ in loop.tape.ops)
    err, sc = max_err(x0, g["ddim50"]), g["ddim50"].abs().max().item()
    print("ViT DDIM-50", mode, "max|d| vs reference:", err, "scale", sc)
    assert err < 5e-4 * max(1.0, sc)


def test_dit_token_forward_matches_oracle(golden):
    from azula_amd.nn import DiT

    g = golden("g5_vit")
    shapes = {k: tuple(v) for k, v in g.meta["shapes"].items()}
    shapes["pos_embedding.2.weight"] = (64, 64)  # pos_channels = 1 for the plain DiT
    sd = synth.synth_state_dict(shapes, 17)
    net = DiT(16, 16, mod_features=32, hid_channels=64, hid_blocks=2, attention_heads=4)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(5)
    x, mod = torch.randn(2, 24, 16, generator=gen), torch.randn(2, 32, generator=gen)
    ref = nets.dit_forward(sd, dict(hid_channels=64, hid_blocks=2, attention_heads=4), x, mod)
    y = net(x.cuda(), mod.cuda())
    assert max_err(y, ref) < 1e-4 * max(1.0, ref.abs().max().item())


def test_vit_ddim50_fused_matches_reference(golden):
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler

    g = golden("g6_vit_loop")
    cfg = g.meta["cfg"]
    w = TimeModulated(build_vit(cfg), cfg["mod_features"], name="vit")
    w.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    den = KarrasDenoiser(w, VPSchedule()).cuda().eval()
    smp = DDIMSampler(den, steps=50, silent=True)
    x0 = smp(g["x1"].cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None
    err, sc = max_err(x0, g["ddim50"]), g["ddim50"].abs().max().item()
    print("ViT DDIM-50 fused max|d| vs reference:", err, "scale", sc)
    assert err < 5e-4 * max(1.0, sc)


@pytest.mark.parametrize("name", ["vit_rope_swiglu", "vit_relu2_noqknorm", "vit_relu", "g24_vit_hd48_rope", "g24_vit_hd96_noqknorm", "g24_vit_hd24"])
def test_vit_variants_match_reference(golden, name):
    """RoPE (attention.py:112-156), SwiGLU / ReLU^2 / ReLU FFNs (layers.py:71-110), qk_norm=False; G24: head sizes 48 (RoPE +
    q/k norm), 96 and 24 -- run zero-padded to 64 / 128 / 32 (engine.ATTN_HEAD_DIMS; azula/nn/attention.py:35-51 takes any)."""
    from azula_amd.nn import ViT

    g = golden(name if name.startswith("g24_") else "g5_" + name)
    cfg = g.meta["cfg"]
    extra = {k: cfg[k] for k in ("rope", "ffn_activation", "qk_norm") if k in cfg}
    net = ViT(cfg["in_channels"], cfg["out_channels"], hid_channels=cfg["hid_channels"], hid_blocks=cfg["hid_blocks"],
              attention_heads=cfg["attention_heads"], patch_size=cfg["patch_size"], mod_features=cfg["mod_features"], **extra)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in g.meta["shapes"].items()}
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    net = net.cuda().eval()
    y = net(g["x"].cuda(), g["modB"].cuda())
    err, sc = max_err(y, g["y_modB"]), g["y_modB"].abs().max().item()
    print(name, "max|d|", err, "scale", sc)
    assert err < 1e-4 * max(1.0, sc)
    if name.startswith("g24_"):
        ops = [a for _, a, n in next(iter(net._plans.values())).tape.ops if n.startswith("az_attention")]
        hd = cfg["hid_channels"] // cfg["attention_heads"]
        assert ops and all(a[0]._obj.norm_dim == hd and a[0]._obj.head_dim > hd for a in ops)  # (padded heads, norm over the real size)
