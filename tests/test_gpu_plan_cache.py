r"""The fused (captured-graph) loop must always see the CURRENT state of the sampler and the denoiser, like the
reference's loop does (``azula/sample.py:139-161`` re-reads weights, guidance and hyper-parameters on every call):
reloaded weights, a changed guidance value (float or 0-d tensor), eta / temperature, train/eval.  Every case compares
the re-used sampler bitwise with a freshly built one and, where a fixture exists, with the reference's golden vector."""

import math

import pytest
import torch

from conftest import max_err
from oracle import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _unet_denoiser(g, seed=None):
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated, UNet
    from azula_amd.noise import VPSchedule

    cfg = g.meta["cfg"]
    net = UNet(cfg["in_channels"], cfg["out_channels"], hid_channels=cfg["hid_channels"], hid_blocks=cfg["hid_blocks"],
               norm=cfg["norm"], groups=cfg["groups"], mod_features=cfg["mod_features"])
    w = TimeModulated(net, cfg["mod_features"], name="unet")
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"] if seed is None else seed)
    w.load_state_dict(sd)
    return KarrasDenoiser(w, VPSchedule()).cuda().eval(), sd


def test_fused_loop_sees_reloaded_weights(golden):
    from azula_amd.sample import DDIMSampler

    g = golden("g6_unet_loop")
    x1 = g["x1"].cuda()
    den, sd_gold = _unet_denoiser(g)
    smp = DDIMSampler(den, steps=64, silent=True)
    x0_gold = smp(x1)
    assert max_err(x0_gold, g["ddim64"]) < 5e-4 * max(1.0, g["ddim64"].abs().max().item())
    loop_a = next(iter(smp._fused_cache.values()))

    # new weights into the SAME denoiser / sampler objects
    den_b, sd_b = _unet_denoiser(g, seed=g.meta["weight_seed"] + 1)
    den.backbone.load_state_dict(sd_b)
    x0_b = smp(x1)
    fresh_b = DDIMSampler(den_b, steps=64, silent=True)(x1)
    assert torch.equal(x0_b, fresh_b), "the cached plan replayed stale packed weights"
    assert not torch.equal(x0_b, x0_gold)
    assert next(iter(smp._fused_cache.values())) is not loop_a and len(smp._fused_cache) == 1

    # and back: the golden trajectory again, bit-identical to the first run
    den.backbone.load_state_dict(sd_gold)
    x0_again = smp(x1)
    assert torch.equal(x0_again, x0_gold)
    assert max_err(x0_again, g["ddim64"]) < 5e-4 * max(1.0, g["ddim64"].abs().max().item())

    # an in-place update of ONE tensor (EMA swap, optimiser step) is seen too
    p = next(den.backbone.unet.descent[0][0].parameters())
    p.mul_(1.5)
    x0_c = smp(x1)
    assert not torch.equal(x0_c, x0_gold)
    p.div_(1.5)

    # unchanged state: the plan is re-used (no rebuild per call)
    smp(x1)
    loop_c = next(iter(smp._fused_cache.values()))
    smp(x1)
    assert next(iter(smp._fused_cache.values())) is loop_c


def test_fused_loop_sees_half_cast(golden):
    from azula_amd.sample import DDIMSampler

    g = golden("g6_unet_loop")
    x1 = g["x1"].cuda()
    den, _ = _unet_denoiser(g)
    smp = DDIMSampler(den, steps=8, silent=True)
    x32 = smp(x1)
    den.backbone.bfloat16()
    x16 = smp(x1)
    den2, _ = _unet_denoiser(g)
    den2.backbone.bfloat16()
    assert torch.equal(x16, DDIMSampler(den2, steps=8, silent=True)(x1))
    assert not torch.equal(x16, x32)


def test_fused_loop_sees_sampler_hyperparameters(golden):
    from azula_amd.sample import DDIMSampler, ItoSampler

    g = golden("g6_unet_loop")
    x1 = g["x1"].cuda()
    den, _ = _unet_denoiser(g)

    smp = ItoSampler(den, steps=8, eta=1.0, temperature=1.0, silent=True)
    torch.manual_seed(5)
    a = smp(x1)
    smp.temperature = 2.0
    torch.manual_seed(5)
    b = smp(x1)
    torch.manual_seed(5)
    fresh = ItoSampler(den, steps=8, eta=1.0, temperature=2.0, silent=True)(x1)
    assert torch.equal(b, fresh) and not torch.equal(a, b)

    smp = DDIMSampler(den, steps=8, eta=0.5, silent=True)
    torch.manual_seed(5)
    a = smp(x1)
    smp.eta = 1.0
    torch.manual_seed(5)
    b = smp(x1)
    torch.manual_seed(5)
    fresh = DDIMSampler(den, steps=8, eta=1.0, silent=True)(x1)
    assert torch.equal(b, fresh) and not torch.equal(a, b)
    smp.eta = 0.0  # the noise stream is no longer read: a different captured kernel
    torch.manual_seed(5)
    c = smp(x1)
    torch.manual_seed(5)
    assert torch.equal(c, DDIMSampler(den, steps=8, eta=0.0, silent=True)(x1))

    smp.start, smp.steps = 0.8, 5
    assert torch.equal(smp(x1), DDIMSampler(den, steps=5, start=0.8, eta=0.0, silent=True)(x1))

    den.schedule.alpha_min = 1e-2  # the schedule object is re-read as well
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.noise import VPSchedule

    den_f = KarrasDenoiser(den.backbone, VPSchedule(alpha_min=1e-2)).cuda().eval()
    assert torch.equal(smp(x1), DDIMSampler(den_f, steps=5, start=0.8, eta=0.0, silent=True)(x1))


def test_invalidate_after_a_raw_write(golden):
    """Writes that do not bump a tensor's version counter (`.data.copy_`) are invisible to the plan key -- documented in
    INTEGRATION.md; `Sampler.invalidate()` is the explicit way to make the next call re-read everything."""
    from azula_amd.sample import DDIMSampler

    g = golden("g6_unet_loop")
    x1 = g["x1"].cuda()
    den, _ = _unet_denoiser(g)
    eta = torch.tensor(0.5, device="cuda")
    smp = DDIMSampler(den, steps=8, eta=eta, silent=True)
    torch.manual_seed(5)
    a = smp(x1)
    eta.data.copy_(torch.tensor(1.0))  # (no version bump: the cached value 0.5 stays in use ...)
    smp.invalidate()                   # (... until the caller says so)
    torch.manual_seed(5)
    b = smp(x1)
    torch.manual_seed(5)
    fresh = DDIMSampler(den, steps=8, eta=torch.tensor(1.0, device="cuda"), silent=True)(x1)
    assert torch.equal(b, fresh) and not torch.equal(a, b)


def test_invalidate_after_a_raw_weight_write(golden):
    """`.data.copy_()` on a convolution weight changes neither `_version` nor `data_ptr()`, so the backbone's
    own plan cache (packed direct / bf16x3 / Winograd-domain filters) would survive a rebuilt sampler loop: `invalidate()` drops
    those caches too, and the next call equals a freshly built denoiser with the new weights."""
    from azula_amd.sample import DDIMSampler

    g = golden("g6_unet_loop")
    x1 = g["x1"].cuda()
    den, _ = _unet_denoiser(g)
    smp = DDIMSampler(den, steps=8, silent=True)
    a = smp(x1)
    convs = [m for m in den.modules() if isinstance(getattr(m, "weight", None), torch.nn.Parameter) and m.weight.ndim == 4 and m.weight.shape[-1] == 3]
    w = convs[len(convs) // 2].weight
    v0, p0 = w._version, w.data_ptr()
    w.data.copy_(w.data * 1.5 + 0.01)
    assert (w._version, w.data_ptr()) == (v0, p0)  # (invisible to every plan key)
    assert torch.equal(smp(x1), a)  # documented: the stale plan is still in use ...
    smp.invalidate()                # ... until the caller says so
    b = smp(x1)
    den2, _ = _unet_denoiser(g)
    den2.load_state_dict(den.state_dict())
    fresh = DDIMSampler(den2, steps=8, silent=True)(x1)
    assert torch.equal(b, fresh) and not torch.equal(a, b)


def test_invalidate_after_a_raw_weight_write_adm():
    from azula_amd.sample import DDIMSampler

    den, _ = _small_cond_adm()
    torch.manual_seed(1)
    x1 = torch.randn(2, 3, 16, 16, device="cuda")
    lab = torch.tensor([1, 2], device="cuda")
    smp = DDIMSampler(den, steps=4, silent=True)
    a = smp(x1, label=lab)
    w = den.backbone.input_blocks[1][0].in_layers[2].weight
    w.data.copy_(w.data * 1.5 + 0.01)
    smp.invalidate()
    b = smp(x1, label=lab)
    den2, _ = _small_cond_adm()
    den2.load_state_dict(den.state_dict())
    fresh = DDIMSampler(den2, steps=4, silent=True)(x1, label=lab)
    assert torch.equal(b, fresh) and not torch.equal(a, b)


def _small_cond_adm():
    from azula_amd.guidance import CFGDenoiser
    from azula_amd.plugins import adm

    torch.manual_seed(0)
    den = adm.make_model(image_size=16, num_channels=32, channel_mult=(1, 2), attention_resolutions=(8,), num_classes=10,
                         num_res_blocks=2, num_head_channels=32, resblock_updown=True, use_scale_shift_norm=True)
    g = torch.Generator().manual_seed(123)
    for _, v in sorted(den.backbone.state_dict().items()):
        if torch.is_floating_point(v) and v.ndim > 1 and not torch.any(v != 0):
            v.copy_(torch.randn(v.shape, generator=g) / math.sqrt(v[0].numel()))
    den = den.cuda().eval()
    return den, CFGDenoiser(den)


def test_cfg_guidance_value_is_reread_every_call():
    from azula_amd.sample import DDIMSampler

    den, cfg = _small_cond_adm()
    torch.manual_seed(1)
    x1 = torch.randn(2, 3, 16, 16, device="cuda")
    pos = {"label": torch.tensor([1, 2], device="cuda")}
    neg = {"label": torch.tensor([0, 0], device="cuda")}
    smp = DDIMSampler(cfg, steps=6, silent=True)
    a = smp(x1, positive=pos, negative=neg, guidance=torch.tensor(1.5))
    loop = next(iter(smp._fused_cache.values()))
    b = smp(x1, positive=pos, negative=neg, guidance=torch.tensor(4.0))
    assert next(iter(smp._fused_cache.values())) is loop, "a guidance change must not rebuild the graph"
    fresh = DDIMSampler(cfg, steps=6, silent=True)(x1, positive=pos, negative=neg, guidance=4.0)
    assert torch.equal(b, fresh) and not torch.equal(a, b)
    c = smp(x1, positive=pos, negative=neg, guidance=1.5)  # float after tensor: same table
    assert torch.equal(c, a)
    # other labels on the same plan
    pos2 = {"label": torch.tensor([7, 3], device="cuda")}
    d = smp(x1, positive=pos2, negative=neg, guidance=1.5)
    assert torch.equal(d, DDIMSampler(cfg, steps=6, silent=True)(x1, positive=pos2, negative=neg, guidance=1.5))
    assert not torch.equal(d, c)


def test_train_eval_switch_changes_the_clip():
    from azula_amd.sample import DDIMSampler

    den, _ = _small_cond_adm()
    assert den.clip_mean
    torch.manual_seed(1)
    x1 = 3 * torch.randn(2, 3, 16, 16, device="cuda")
    lab = torch.tensor([1, 2], device="cuda")
    smp = DDIMSampler(den, steps=4, silent=True)
    clipped = smp(x1, label=lab)
    den.train()  # reference: the mean is clipped in eval mode only (adm/__init__.py:131-134)
    unclipped = smp(x1, label=lab)
    den.eval()
    again = smp(x1, label=lab)
    assert torch.equal(again, clipped) and not torch.equal(unclipped, clipped)


def test_transition_clip_propagates_nan():
    import ctypes as C

    from azula_amd import _lib
    from azula_amd.engine import transition_args

    n = 4096 + 3
    x = torch.randn(n, device="cuda")
    F = torch.randn(n, device="cuda")
    F[5], F[4097] = float("nan"), float("nan")
    out, mean = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    row = torch.zeros(_lib.COEF_WORDS, device="cuda")
    col = {k: i for i, k in enumerate(_lib.COEF_FIELDS)}
    row[col["c_skip"]], row[col["c_out"]], row[col["alpha_s"]], row[col["k_x"]] = 0.5, 0.7, 0.9, 0.3
    row[col["clip_lo"]], row[col["clip_hi"]] = -1.0, 1.0
    a = transition_args(x_t=x.data_ptr(), F=F.data_ptr(), x_s=out.data_ptr(), mean_out=mean.data_ptr(), batch=1, channels=1,
                        inner=n, f_channels=1, coef=row.data_ptr())
    _lib.call("az_transition_f32", C.byref(a), _lib.stream_ptr())
    ref = torch.clip(0.5 * x + 0.7 * F, -1.0, 1.0)
    assert torch.isnan(mean[5]) and torch.isnan(mean[4097]) and torch.isnan(out[5])
    assert torch.equal(torch.isnan(mean), torch.isnan(ref))
    ok = ~torch.isnan(ref)
    assert torch.equal(mean[ok], ref[ok])


def test_tensor_valued_hyper_parameters_reach_the_table(golden):
    """eta held as a 0-d TENSOR (and a schedule scalar held as a tensor) changed between two calls of the same sampler: the
    captured loop must re-upload its coefficient table, like the reference re-reads them."""
    from azula_amd.sample import DDIMSampler

    g = golden("g6_unet_loop")
    x1 = g["x1"].cuda()
    den, _ = _unet_denoiser(g)
    smp = DDIMSampler(den, steps=6, eta=torch.tensor(0.25), silent=True)
    torch.manual_seed(3)
    a = smp(x1)
    smp.eta = torch.tensor(1.0)
    torch.manual_seed(3)
    b = smp(x1)
    torch.manual_seed(3)
    fresh = DDIMSampler(den, steps=6, eta=1.0, silent=True)(x1)
    assert len(smp._fused_cache) == 1 and next(iter(smp._fused_cache.values())).graph is not None
    assert torch.equal(b, fresh) and not torch.equal(a, b)
    # a schedule scalar as a tensor attribute
    den.schedule.alpha_min = torch.tensor(1e-2, dtype=torch.float64)
    torch.manual_seed(3)
    c = smp(x1)
    den2, _ = _unet_denoiser(g)
    den2.schedule.alpha_min = 1e-2
    torch.manual_seed(3)
    fresh2 = DDIMSampler(den2, steps=6, eta=1.0, silent=True)(x1)
    assert max_err(c, fresh2) < 1e-6 * max(1.0, fresh2.abs().max().item()) and not torch.equal(c, b)
