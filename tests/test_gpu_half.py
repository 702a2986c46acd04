r"""Half-precision modules on the GPU: a backbone cast with ``.bfloat16()`` / ``.half()`` runs its
convolutions and token GEMMs on the bf16 / f16 MFMA kernel (``az_conv2d_{bf16,f16}_f32``, fp32 accumulation, fp32
activations).  Bar: the reference's own mixed-precision tolerance (tests/test_nn_unet.py:78-91: q99 < 1e-3, max < 1e-2
between the half and the fp32 forward on O(1) outputs), scaled by the output magnitude; bf16 has 3 fewer mantissa
bits than f16, so its bound is 8x."""

import pytest
import torch

from conftest import max_err
from oracle import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
HALVES = [torch.float16, torch.bfloat16]


def _bar(y32, y16, half):
    err = (y32.float() - y16.float()).abs().flatten()
    scale = max(1.0, y32.abs().max().item())
    k = 1.0 if half == torch.float16 else 8.0
    return torch.quantile(err, 0.99).item() / scale, err.max().item() / scale, 1e-3 * k, 1e-2 * k


def _uses_half_kernel(plan, half):
    want = "az_conv2d_f16_f32" if half == torch.float16 else "az_conv2d_bf16_f32"
    names = {n for _, _, n in plan.tape.ops}
    return want in names and "az_conv2d_f32" not in names and "az_conv2d_winograd_f32" not in names and "az_conv2d_winograd_x3_f32" not in names


@pytest.mark.parametrize("half", HALVES)
def test_vit_half(golden, half):
    from test_gpu_vit import build_vit

    g = golden("g5_vit")
    net = build_vit(g.meta["cfg"])
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    net = net.cuda().eval().to(half)
    y = net(g["x"].cuda().to(half), g["modB"].cuda().to(half))
    assert y.dtype == half and _uses_half_kernel(next(iter(net._plans.values())), half)
    q99, mx, bq, bm = _bar(g["y_modB"].cuda(), y, half)
    print("vit", half, "q99/scale", q99, "max/scale", mx)
    assert q99 < bq and mx < bm


@pytest.mark.parametrize("half", HALVES)
def test_adm_half(golden, half):
    from test_gpu_adm import build

    g = golden("g5_adm_cond_neworder")
    den, _, _ = build(g)
    den.backbone.to(half)
    out = den.backbone(g["x"].cuda().to(half), g["idx"].cuda(), y=g["y"].cuda())
    assert out.dtype == half and _uses_half_kernel(next(iter(den.backbone._plans.values())), half)
    q99, mx, bq, bm = _bar(g["out"].cuda(), out, half)
    print("adm", half, "q99/scale", q99, "max/scale", mx)
    # the reference has no ADM test; its UNet bar is applied with a factor 2 (30 GroupNorm / FiLM layers deep, and the
    # fp16 rounding of the OUTPUT alone is 2^-11 relative = half the q99 bar)
    assert q99 < 2 * bq and mx < 2 * bm
    # through the denoiser: x_t stays fp32, the backbone input is rounded to the module dtype (reference semantics)
    mean = den(g["x"].cuda(), torch.tensor(0.7, device="cuda"), label=g["y"].cuda()).mean
    assert mean.dtype == torch.float32
    assert max_err(mean, g["mean_t07"]) < (5e-2 if half == torch.float16 else 2e-1)


@pytest.mark.parametrize("half", HALVES)
def test_jit_half_fused_sampling(golden, half):
    from azula_amd.sample import DDIMSampler
    from test_gpu_jit import build

    g = golden("g10_jit_ctx")
    den = build(g)
    x1, y = g["x1"].cuda(), g["y"].long().cuda()
    ref = DDIMSampler(den, steps=8, silent=True)(x1, label=y)
    den.backbone.to(half)
    smp = DDIMSampler(den, steps=8, silent=True)
    x0 = smp(x1, label=y)
    loop = next(iter(smp._fused_cache.values()))
    assert loop.graph is not None and x0.dtype == torch.float32
    names = {n for _, _, n in loop.tape.ops}
    assert ("az_conv2d_f16_f32" if half == torch.float16 else "az_conv2d_bf16_f32") in names
    rel = (x0 - ref).abs().max().item() / ref.abs().max().item()
    print("jit DDIM-8", half, "rel to fp32 weights", rel)
    assert rel < (2e-2 if half == torch.float16 else 1e-1)


@pytest.mark.parametrize("half", HALVES)
@pytest.mark.parametrize("hd,T,rms,rope", [(64, 256, True, False), (16, 77, False, False), (80, 148, True, True), (32, 130, True, True),
                                          (128, 64, False, False)])
def test_attention_half_kernel(half, hd, T, rms, rope):
    """az_attention_{bf16,f16}_f32 against fp32 SDPA on the same (fp32) q, k, v: '(n H C)' fused-QKV layout, ragged token
    counts, q/k RMS norm with gains, 2-D rotary tables.  Error budget = operand rounding of q, k, v and P."""
    import math

    from azula_amd.engine import Builder
    from oracle import nets

    g = torch.Generator().manual_seed(hd + T)
    B, heads = 2, 3
    Cc = heads * hd
    qkv = torch.randn(B, T, 3 * Cc, generator=g)
    qw, kw = 1 + 0.2 * torch.randn(hd, generator=g), 1 + 0.2 * torch.randn(hd, generator=g)
    bld = Builder(torch.device("cuda"), half=half)
    qa = bld.new_act(B, T, 1, 3 * Cc, pinned=True)
    qa.buf[: qkv.numel()].copy_(qkv.reshape(-1).cuda())
    rt = None
    if rope:
        ang = torch.rand(T, heads, hd // 2, generator=g) * 6.0
        rt = (bld.const(torch.cos(ang)), bld.const(torch.sin(ang)))
    out = bld.attention(qa, heads, "3HC", rms, 1 / math.sqrt(hd), eps=1e-6, rope=rt,
                        qk_weight=(bld.const(qw), bld.const(kw)) if rms else None)
    assert bld.tape.ops[-1][2] == ("az_attention_f16_f32" if half == torch.float16 else "az_attention_bf16_f32")
    bld.finish()
    bld.tape.run()
    q, k, v = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    if rms:
        q, k = nets.jit_rms_norm(q, qw), nets.jit_rms_norm(k, kw)
    if rope:
        cos = torch.cos(ang).permute(1, 0, 2).repeat_interleave(2, dim=-1)  # (heads, T, hd): pairs share an angle
        sin = torch.sin(ang).permute(1, 0, 2).repeat_interleave(2, dim=-1)
        q, k = nets.jit_rotate(q, cos, sin), nets.jit_rotate(k, cos, sin)
    want = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, Cc)
    err = max_err(out.buf[: want.numel()].view(B, T, Cc), want)
    tol = (4e-3 if half == torch.float16 else 3e-2) * max(1.0, want.abs().max().item())
    print(half, hd, T, "max|d|", err, "scale", want.abs().max().item())
    assert err < tol
