r"""JiT plugin on the GPU: the compiled backbone, JITDenoiser, fused DDIM and CFG loops
against reference-generated vectors (G10), plus the kernels this path added to the C ABI."""

import ctypes as C
import math

import pytest
import torch

from conftest import max_err
from oracle import nets, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
CONFIGS = ("jit_ctx", "jit_noctx_hd32", "jit_hd80", "g24_jit_hd48")  # (G24: heads of 48 channels, zero-padded to 64)


@pytest.fixture(scope="module")
def az():
    from azula_amd import _lib

    _lib.lib()
    return _lib


def build(g):
    from azula_amd.plugins import jit

    cfg = g.meta["cfg"]
    net = jit.JiT(**cfg)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    return jit.JITDenoiser(net, num_classes=cfg["num_classes"]).cuda().eval()


@pytest.mark.parametrize("name", CONFIGS)
def test_jit_backbone_matches_reference(golden, name):
    g = golden(name if name.startswith("g24_") else "g10_" + name)
    den = build(g)
    x, t, y = g["x"].cuda(), g["t"].cuda(), g["y"].long().cuda()
    sc = g["out"].abs().max().item()
    out = den.backbone(x, t, y)
    print(name, "per-sample t max|d|", max_err(out, g["out"]), "scale", sc)
    assert max_err(out, g["out"]) < 2e-5 * sc
    out = den.backbone(x, t[:1], y)
    assert max_err(out, g["out_shared"]) < 2e-5 * sc
    assert torch.equal(out, den.backbone(x, t[:1], y))  # plan replay is deterministic


@pytest.mark.parametrize("name", CONFIGS)
def test_jit_denoiser_and_fused_loops(golden, name):
    from azula_amd.guidance.cfg import CFGDenoiser
    from azula_amd.sample import DDIMSampler

    g = golden(name if name.startswith("g24_") else "g10_" + name)
    den = build(g)
    x, y = g["x"].cuda(), g["y"].long().cuda()
    sc = g["mean_t04"].abs().max().item()
    assert max_err(den(x, torch.tensor(0.4, device="cuda"), label=y).mean, g["mean_t04"]) < 2e-4 * sc
    assert max_err(den(x, torch.tensor(0.7, device="cuda")).mean, g["mean_null"]) < 2e-4 * sc  # null class

    x1 = g["x1"].cuda()
    smp = DDIMSampler(den, steps=8, silent=True)
    x0 = smp(x1, label=y)
    assert next(iter(smp._fused_cache.values())).graph is not None, "JiT must run inside the captured step graph"
    sc = max(1.0, g["ddim8"].abs().max().item())
    print(name, "ddim8 max|d|", max_err(x0, g["ddim8"]), "scale", sc)
    assert max_err(x0, g["ddim8"]) < 1e-4 * sc
    # new labels reuse the captured graph (prepare() refreshes the label buffer)
    y2 = torch.roll(y, 1)
    assert not torch.equal(smp(x1, label=y2), x0)
    assert torch.equal(smp(x1, label=y), x0)

    cfg = DDIMSampler(CFGDenoiser(den), steps=6, silent=True)
    x0 = cfg(x1, positive={"label": y}, guidance=g.meta["guidance"])
    assert next(iter(cfg._fused_cache.values())).graph is not None
    sc = max(1.0, g["cfg_ddim6"].abs().max().item())
    print(name, "cfg ddim6 max|d|", max_err(x0, g["cfg_ddim6"]), "scale", sc)
    assert max_err(x0, g["cfg_ddim6"]) < 2e-4 * sc


# ---------------------------------------------------------------------------------------- kernels
def test_token_window_kernels(az):
    B, L, Lc, cs = 3, 10, 4, 24
    g = torch.Generator().manual_seed(0)
    x, row, pos = torch.randn(B, L, cs, generator=g), torch.randn(B, cs, generator=g), torch.randn(Lc, cs, generator=g)
    dx, drow, dpos = x.cuda(), row.cuda(), pos.cuda()
    wide = torch.full((B, L + Lc, cs), 7.0, device="cuda")
    s = az.stream_ptr()
    az.call("az_token_fill_f32", wide.data_ptr(), L + Lc, 0, Lc, drow.data_ptr(), cs, dpos.data_ptr(), B, cs, s)
    az.call("az_token_copy_f32", wide.data_ptr(), L + Lc, Lc, dx.data_ptr(), L, 0, L, B, cs, s)
    want = torch.cat((row[:, None, :] + pos[None], x), dim=1)
    assert torch.equal(wide.cpu(), want)
    body = torch.empty(B, L - 2, cs, device="cuda")
    az.call("az_token_copy_f32", body.data_ptr(), L - 2, 0, wide.data_ptr(), L + Lc, Lc + 1, L - 2, B, cs, s)
    assert torch.equal(body.cpu(), x[:, 1:-1])
    with pytest.raises(az.AzulaAmdError):  # window past the end of the source
        az.call("az_token_copy_f32", body.data_ptr(), L - 2, 0, wide.data_ptr(), L + Lc, Lc + 4, L - 2, B, cs, s)


@pytest.mark.parametrize("shared", [False, True])
def test_timestep_embedding(az, shared):
    rows, dim = 5, 256
    t = torch.rand(1 if shared else rows)
    dt = t.cuda()
    out = torch.empty(rows, dim, device="cuda")
    az.call("az_timestep_embedding_f32", out.data_ptr(), dim, dt.data_ptr(), 0 if shared else 1, rows, dim // 2, 10000.0,
            az.stream_ptr())
    want = nets.jit_timestep_embedding(t.expand(rows), dim)
    assert max_err(out, want) < 5e-7  # device expf / sincosf vs the host libm: a couple of ulp of values <= 1


def test_rownorm_with_gain(az):
    from azula_amd.engine import Builder

    g = torch.Generator().manual_seed(1)
    B, L, Cc = 2, 9, 64
    x = torch.randn(B, L, Cc, generator=g)
    w = 1 + 0.1 * torch.randn(Cc, generator=g)
    mod = torch.randn(B, 2 * Cc, generator=g)
    bld = Builder(torch.device("cuda"))
    xa = bld.new_act(B, L, 1, Cc, pinned=True)
    xa.buf[: x.numel()].copy_(x.reshape(-1).cuda())
    dm = bld.const(mod)
    y = bld.row_norm(xa, 1, weight=bld.const(w), scale=dm, shift=dm, scale_off=Cc, shift_off=0, bstride=2 * Cc, eps=1e-6)
    bld.finish()
    bld.tape.run()
    want = nets.jit_rms_norm(x, w) * (1 + mod[:, None, Cc:]) + mod[:, None, :Cc]
    assert max_err(y.buf[: x.numel()].view(B, L, Cc), want) < 1e-5


@pytest.mark.parametrize("hd", [16, 64, 80])
def test_attention_with_gains_and_2d_rope(az, hd):
    """'(3 H C)' fused QKV + weighted q/k RMSNorm + JiT's 2-D rotary tables + context tokens."""
    from azula_amd.engine import Builder
    from azula_amd.plugins.jit.model import rotary_tables

    g = torch.Generator().manual_seed(hd)
    B, heads, grid, ctx = 2, 3, 4, 5
    L, Cc = ctx + grid * grid, heads * hd
    qkv = torch.randn(B, L, 3 * Cc, generator=g)
    qw, kw = 1 + 0.2 * torch.randn(hd, generator=g), 1 + 0.2 * torch.randn(hd, generator=g)
    bld = Builder(torch.device("cuda"))
    qa = bld.new_act(B, L, 1, 3 * Cc, pinned=True)
    qa.buf[: qkv.numel()].copy_(qkv.reshape(-1).cuda())
    rope = tuple(bld.const(t) for t in rotary_tables(hd, heads, grid, ctx))
    out = bld.attention(qa, heads, "3HC", True, 1 / math.sqrt(hd), eps=1e-6, rope=rope, qk_weight=(bld.const(qw), bld.const(kw)))
    bld.finish()
    bld.tape.run()
    q, k, v = qkv.reshape(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    cos, sin = nets.jit_rope_tables(hd, grid, ctx)
    q = nets.jit_rotate(nets.jit_rms_norm(q, qw), cos, sin)
    k = nets.jit_rotate(nets.jit_rms_norm(k, kw), cos, sin)
    want = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, L, Cc)
    assert max_err(out.buf[: want.numel()].view(B, L, Cc), want) < 2e-5
