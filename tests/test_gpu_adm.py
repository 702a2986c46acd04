r"""ADM plugin (AblatedDenoiser + guided-diffusion UNet) and classifier-free guidance on the GPU,
against the reference-generated golden vectors (G5/G6) and the oracle."""

import pytest
import torch

from conftest import max_err
from oracle import nets, sampling, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

NAMES = ["adm_uncond", "adm_cond_neworder"]


def build(g):
    from azula_amd.plugins import adm

    cfg = g.meta["cfg"]
    den = adm.make_model(**cfg)
    shapes = {k: tuple(v) for k, v in g.meta["shapes"].items()}
    assert {k: tuple(v.shape) for k, v in den.backbone.state_dict().items()} == shapes
    sd = synth.synth_state_dict(shapes, g.meta["weight_seed"])
    den.backbone.load_state_dict(sd)
    return den.cuda().eval(), sd, cfg


@pytest.mark.parametrize("name", NAMES)
def test_adm_backbone_matches_reference(golden, name):
    g = golden("g5_" + name)
    den, _, _ = build(g)
    y = g["y"].cuda() if "y" in g else None
    out = den.backbone(g["x"].cuda(), g["idx"].cuda(), y=y)
    err, sc = max_err(out, g["out"]), g["out"].abs().max().item()
    print(name, "backbone max|d|", err, "scale", sc)
    assert err < 7e-6 * max(1.0, sc)  # measured 3.6e-6 / 2.7e-6 on scale 2.9 / 2.1 (MI355X, round 5): bound = 5.6 x


@pytest.mark.parametrize("name", NAMES)
def test_adm_posterior_matches_reference(golden, name):
    g = golden("g5_" + name)
    den, _, _ = build(g)
    kw = {"label": g["y"].cuda()} if "y" in g else {}
    q = den(g["x"].cuda(), torch.tensor(0.7, device="cuda"), **kw)
    em, ev = max_err(q.mean, g["mean_t07"]), max_err(q.var, g["var_t07"])
    print(name, "posterior mean / var max|d|", em, ev)
    assert em < 5e-5
    assert q.mean.abs().max() <= 1.0  # clipped in eval mode
    assert ev < 5e-5 * max(1.0, g["var_t07"].abs().max().item())


@pytest.mark.parametrize("name", NAMES)
def test_adm_ddim16_fused_matches_reference(golden, name):
    from azula_amd.sample import DDIMSampler

    g = golden("g5_" + name)
    den, _, _ = build(g)
    kw = {"label": g["y"].cuda()} if "y" in g else {}
    smp = DDIMSampler(den, steps=16, silent=True)
    x0 = smp(g["x1"].cuda(), **kw)
    assert next(iter(smp._fused_cache.values())).graph is not None, "fused path not taken"
    err = max_err(x0, g["ddim16"])
    print(name, "DDIM-16 max|d|", err)
    # |x0| <= ~1 (means are clipped to [-1, 1]); c_out reaches -100 at t = 1.  Measured 1.2e-5 / 1.9e-5: bound = 5 x
    assert err < 1e-4


def test_adm_ddpm8_device_rng_matches_oracle(golden):
    from azula_amd.sample import DDPMSampler

    g = golden("g5_adm_uncond")
    den, sd, cfg = build(g)
    x1 = g["x1"].cuda()
    torch.manual_seed(9)
    eps = [torch.randn_like(x1).cpu() for _ in range(8)]
    torch.manual_seed(9)
    x0 = DDPMSampler(den, steps=8, silent=True)(x1)
    sig = sampling.adm_sigmas(cfg["discrete_schedule"], cfg["discrete_steps"])
    bb = lambda a, i, y=None: nets.adm_unet_forward(sd, cfg, a, i, y)  # noqa: E731
    omean = lambda xx, t: sampling.adm_posterior(bb, xx, t, sig)[0]  # noqa: E731
    ref = sampling.sample(omean, g["x1"], schedule=lambda t: sampling.vp_schedule(t, 1e-2, 1e-2), steps=8, eta=None, eps_list=eps)
    err = max_err(x0, ref)
    print("ADM DDPM-8 max|d| vs oracle", err)
    assert err < 6e-5  # measured 1.2e-5: bound = 5 x


def _adm_oracle(g, sd, cfg):
    sig = sampling.adm_sigmas(cfg["discrete_schedule"], cfg["discrete_steps"])
    bb = lambda a, i, y=None: nets.adm_unet_forward(sd, cfg, a, i, y)  # noqa: E731
    omean = lambda xx, t: sampling.adm_posterior(bb, xx, t, sig)[0]  # noqa: E731
    return omean, (lambda t: sampling.vp_schedule(t, 1e-2, 1e-2))


def test_adm_ddim64_full_length_matches_oracle(golden):
    """BASELINE configs[4]'s trajectory LENGTH (DDIM-64) on a conv + attention backbone, captured graph, against the oracle's
    loop (only 16-step ADM trajectories were pinned).  c_out = -100 at t = 1, means clipped to +-1."""
    from azula_amd.sample import DDIMSampler

    g = golden("g5_adm_uncond")
    den, sd, cfg = build(g)
    omean, sched = _adm_oracle(g, sd, cfg)
    smp = DDIMSampler(den, steps=64, silent=True)
    x0 = smp(g["x1"].cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None
    ref = sampling.sample(omean, g["x1"], schedule=sched, steps=64, eta=0.0)
    err = max_err(x0, ref)
    print("ADM DDIM-64 max|d| vs oracle", err, "scale", ref.abs().max().item())
    assert err < 3.5e-5  # measured 7.0e-6 on |x0| <= 1.04 (MI355X, round 4): bound = 5 x


def test_adm_ddpm1000_full_length_matches_oracle(golden):
    """BASELINE configs[3]'s sampler at its real length: DDPM-1000 through the captured graph with the device generator's
    noise, the same 1000 draws copied to the oracle's loop (~20 s of CPU).  This is where fp32 round-off would compound:
    1000 transitions, c_out = -100 near t = 1, fresh noise every step."""
    from azula_amd.sample import DDPMSampler

    g = golden("g5_adm_uncond")
    den, sd, cfg = build(g)
    omean, sched = _adm_oracle(g, sd, cfg)
    x1 = g["x1"].cuda()
    torch.manual_seed(1000)
    eps = [torch.randn_like(x1).cpu() for _ in range(1000)]
    torch.manual_seed(1000)
    smp = DDPMSampler(den, steps=1000, silent=True)
    x0 = smp(x1)
    assert next(iter(smp._fused_cache.values())).graph is not None
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = sampling.sample(omean, g["x1"], schedule=sched, steps=1000, eta=None, eps_list=eps)
    err = max_err(x0, ref)
    rms = (x0.cpu() - ref).pow(2).mean().sqrt().item()
    print("ADM DDPM-1000 max|d| vs oracle", err, "rms", rms, "scale", ref.abs().max().item())
    assert torch.isfinite(x0).all()
    assert err < 2.5e-5 and rms < 2.5e-6  # measured 4.9e-6 / 4.8e-7 (MI355X, round 4): 1000 steps do NOT compound -- bounds = 5 x


def test_cfg_ddim16_fused_and_generic(golden):
    from azula_amd.guidance import CFGDenoiser
    from azula_amd.sample import DDIMSampler

    g = golden("g5_adm_cond_neworder")
    den, _, _ = build(g)
    cfgden = CFGDenoiser(den)
    kwargs = dict(positive={"label": g["y"].cuda()}, negative={"label": g["neg_label"].cuda()}, guidance=2.0)
    smp = DDIMSampler(cfgden, steps=16, silent=True)
    x0 = smp(g["x1"].cuda(), **kwargs)
    ent = next(iter(smp._fused_cache.values()))
    assert ent.graph is not None and len(ent.fused.programs) == 2
    err = max_err(x0, g["cfg_ddim16"])
    print("CFG DDIM-16 fused max|d|", err)
    assert err < 1e-3  # measured 2.4e-4 (guidance 2 triples the difference of two evaluations): bound = 4 x

    class Loop(DDIMSampler):  # generic path: two denoiser calls + az_cfg_combine per step
        def step(self, x_t, t, s, **kw):
            return super().step(x_t, t, s, **kw)

    x0g = Loop(cfgden, steps=16, silent=True)(g["x1"].cuda(), **kwargs)
    # the generic path evaluates the schedule with device libm (as the reference would on a GPU); at
    # t = 1 the ADM preconditioning has c_out = -100, so last-ulp scalar differences are amplified
    print("CFG generic vs fused", max_err(x0g, x0), "generic vs reference", max_err(x0g, g["cfg_ddim16"]))
    # measured 5.5e-5 .. 3.5e-4 (with AZ_STEM_PLANAR=0) / 1.1e-4 .. 2.0e-4: two roundings of the same chaotic amplification
    assert max_err(x0g, x0) < 6e-4 and max_err(x0g, g["cfg_ddim16"]) < 4e-4  # (generic vs reference: measured 7.7e-5)
    # schedule passes through the wrapper (reference cfg.py:31-33)
    assert cfgden.schedule is den.schedule


@pytest.mark.parametrize("name", ["adm_plain_conv", "adm_plain_pool", "adm_film_noupdown", "g24_adm_hd24_legacy", "g24_adm_hd48_hd96_neworder"])
def test_adm_options_outside_the_cards(golden, name):
    """guided-diffusion's default wiring -- ``use_scale_shift_norm=False`` (h + emb), ``resblock_updown=False``
    (Downsample / Upsample layers, with and without ``conv_resample``) -- against the reference's outputs (G14); G24: attention
    heads of 24 (legacy q | k | v order) and of 48 / 96 channels (new order), run zero-padded to 32 / 64 / 128."""
    from azula_amd.sample import DDIMSampler

    g = golden(name if name.startswith("g24_") else "g14_" + name)
    den, _, cfg = build(g)
    y = g["y"].cuda() if "y" in g else None
    out = den.backbone(g["x"].cuda(), g["idx"].cuda(), y=y)
    err, sc = max_err(out, g["out"]), g["out"].abs().max().item()
    kw = {"label": y} if y is not None else {}
    smp = DDIMSampler(den, steps=8, silent=True)
    x0 = smp(g["x1"].cuda(), **kw)
    assert next(iter(smp._fused_cache.values())).graph is not None
    e2 = max_err(x0, g["ddim8"])
    print(name, "backbone max|d|", err, "scale", sc, "DDIM-8", e2, "scale", g["ddim8"].abs().max().item())
    assert err < 5e-6 * max(1.0, sc)  # measured 2.1e-6 .. 2.5e-6 on scale 2.2 .. 2.7: bound = 5 x
    assert e2 < 2.5e-4  # measured 2.8e-5 .. 5.0e-5 (means clipped to +-1, c_out = -100 at t = 1)


@pytest.mark.parametrize("name", ["adm_1d_film_updown", "adm_1d_plain_conv", "adm_1d_plain_pool"])
def test_adm_on_one_dimensional_signals(golden, name):
    """``UNetModel(dims=1)`` (conv_nd / avg_pool_nd, plugins/adm/_src/nn.py:50-77): (B, C, L) signals run as one-row images --
    Conv1d filters in the middle row of 3x3 ones, stride 2 / nearest x2 / the average pool along the width alone (G22)."""
    from azula_amd.sample import DDIMSampler

    g = golden("g22_" + name)
    den, _, cfg = build(g)
    assert g["x"].ndim == 3
    y = g["y"].cuda() if "y" in g else None
    out = den.backbone(g["x"].cuda(), g["idx"].cuda(), y=y)
    assert out.shape == g["out"].shape
    err, sc = max_err(out, g["out"]), g["out"].abs().max().item()
    kw = {"label": y} if y is not None else {}
    smp = DDIMSampler(den, steps=8, silent=True)
    x0 = smp(g["x1"].cuda(), **kw)
    assert x0.shape == g["ddim8"].shape
    assert next(iter(smp._fused_cache.values())).graph is not None  # the captured loop, not the generic one
    e2 = max_err(x0, g["ddim8"])
    print(name, "backbone max|d|", err, "scale", sc, "DDIM-8", e2, "scale", g["ddim8"].abs().max().item())
    assert err < 7e-6 * max(1.0, sc)  # measured 2.0e-6 .. 2.8e-6 on scale 1.6 .. 2.0
    assert e2 < 1.4e-4  # measured 2.8e-6 .. 2.7e-5
    # the posterior (generic call path) on the same signal
    post = den(g["x1"].cuda(), torch.tensor(0.5, device="cuda"), **kw)
    assert post.mean.shape == g["x1"].shape and torch.isfinite(post.mean).all()


@pytest.mark.parametrize("name", ["adm_3d_film_updown", "adm_3d_plain_conv", "adm_3d_plain_pool"])
def test_adm_on_volumes(golden, name):
    """``UNetModel(dims=3)`` (Conv3d / AvgPool3d, plugins/adm/_src/nn.py:9-39; Upsample / Downsample on the inner two axes only,
    _src/unet.py:103-104,128): a (B, C, D, H, W) volume runs as B D planes, every Conv3d as three depth-tap launches of the 2-D
    kernels accumulating in place, the norms over (D H) x W images, attention over D H W tokens (G23: odd and even depths,
    H != W, FiLM + resblock_updown / h + emb with convolutional / pooling resampling)."""
    from azula_amd.sample import DDIMSampler

    g = golden("g23_" + name)
    den, _, cfg = build(g)
    assert g["x"].ndim == 5
    y = g["y"].cuda() if "y" in g else None
    out = den.backbone(g["x"].cuda(), g["idx"].cuda(), y=y)
    assert out.shape == g["out"].shape
    err, sc = max_err(out, g["out"]), g["out"].abs().max().item()
    kw = {"label": y} if y is not None else {}
    smp = DDIMSampler(den, steps=8, silent=True)
    x0 = smp(g["x1"].cuda(), **kw)
    assert x0.shape == g["ddim8"].shape
    assert next(iter(smp._fused_cache.values())).graph is not None  # the captured loop, not the generic one
    e2 = max_err(x0, g["ddim8"])
    print(name, "backbone max|d|", err, "scale", sc, "DDIM-8", e2, "scale", g["ddim8"].abs().max().item())
    assert err < 1e-5 * max(1.0, sc)  # measured 2.5e-6 .. 4.2e-6 on scale 2.1 .. 3.0
    assert e2 < 2e-4  # measured 1.3e-5 .. 6.9e-5 (means clipped to +-1, c_out = -100 at t = 1)
    post = den(g["x1"].cuda(), torch.tensor(0.5, device="cuda"), **kw)
    assert post.mean.shape == g["x1"].shape and torch.isfinite(post.mean).all()


@pytest.mark.parametrize("name", ["adm_3d_plain_conv", "adm_3d_film_updown"])
def test_adm_on_a_single_plane_volume(golden, name):
    """a (B, C, 1, H, W) volume -- the outer depth taps of every Conv3d(padding=1) read only padding (reference:
    torch.nn.Conv3d accepts it, plugins/adm/_src/nn.py:9-39).  The build launches the centre tap only (an out-of-volume tap is
    rejected by the C ABI); checked against the oracle on the same weights."""
    g = golden("g23_" + name)
    den, sd, cfg = build(g)
    x = g["x"][:, :, :1].contiguous()
    assert x.ndim == 5 and x.shape[2] == 1
    y = g["y"] if "y" in g else None
    ref = nets.adm_unet_forward(sd, cfg, x, g["idx"], y)
    out = den.backbone(x.cuda(), g["idx"].cuda(), y=None if y is None else y.cuda())
    assert out.shape == ref.shape
    err, sc = max_err(out, ref), ref.abs().max().item()
    print(name, "depth 1: backbone max|d|", err, "scale", sc)
    assert err < 1e-5 * max(1.0, sc)


@pytest.mark.parametrize("name", NAMES)
def test_adm_fractional_timesteps(golden, name):
    """UNetModel.forward with FRACTIONAL timesteps (plugins/adm/_src/nn.py:90-108): the sinusoid is evaluated on the device
    (az_timestep_embedding_f32) instead of gathered from the integer table; per-sample and shared times (G19)."""
    from azula_amd.plugins import adm

    g = golden("g19_adm_fractional")
    den = adm.make_model(**g.meta[name + "_cfg"])
    den.backbone.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta[name + "_shapes"].items()}, 9))
    net = den.backbone.cuda().eval()
    y = g[name + "_y"].cuda() if name + "_y" in g else None
    out = net(g[name + "_x"].cuda(), g[name + "_t"].cuda(), y=y)
    sc = max(1.0, g[name + "_out"].abs().max().item())
    e = max_err(out, g[name + "_out"])
    out1 = net(g[name + "_x"].cuda(), torch.tensor([417.75], device="cuda"), y=y)
    e1 = max_err(out1, g[name + "_out_shared"])
    print(name, "fractional timesteps max|d|", e, e1, "scale", sc)
    assert e < 7e-6 * sc and e1 < 7e-6 * sc  # measured 3.2e-6 .. 3.8e-6 on scale 2.4 .. 2.8: bound = 5 x
    # the integer path still takes the table
    ops = [n for _, _, n in net.plan(2, g[name + "_x"].shape[2], g[name + "_x"].shape[3], 2, torch.device("cuda", 0)).tape.ops]
    assert "az_timestep_embedding_f32" not in ops


@pytest.mark.parametrize("case", ["ddim", "ddpm", "cfg", "volume", "zab"])
def test_adm_with_an_fp64_clock_runs_as_a_captured_loop(golden, case, monkeypatch):
    """``Sampler(dtype=float64)`` around an ADM denoiser (azula/sample.py:69-94 applies to every denoiser): the captured fp64 loop
    with the posterior mean from the first 3 of the backbone's 6 channels (one az_axpby_f64 per sample), clipped to +-1 by
    az_transition_f64 under the clamp row, CFG as pos + g (pos - neg) of the two clipped means, a channels-last backbone input
    (dims = 3) through an fp32 staging tensor -- against the per-statement fp64 loop (same kernels, same generator draws)."""
    from azula_amd import sample as S
    from azula_amd.guidance.cfg import CFGDenoiser

    name = {"cfg": "g5_adm_cond_neworder", "volume": "g23_adm_3d_plain_conv"}.get(case, "g5_adm_uncond")
    g = golden(name)
    den, _, cfg = build(g)
    x1 = g["x1"].cuda()
    kw = {}
    if case == "cfg":
        # (two sequential programs: the 2B-batch form of the captured CFG step runs its small maps under other tile plans than
        #  two B-batch calls, 1e-4 after c_out = -100 and the guidance -- the fp32 test above measures the same)
        monkeypatch.setenv("AZ_CFG_BATCHED", "0")
        y = g["y"].cuda()
        den, kw = CFGDenoiser(den), dict(positive={"label": y}, negative={"label": torch.zeros_like(y)}, guidance=2.0)
    make = {
        "ddpm": lambda: S.DDPMSampler(den, steps=6, silent=True, dtype=torch.float64),
        "zab": lambda: S.zABSampler(den, order=2, steps=5, silent=True, dtype=torch.float64),
    }.get(case, lambda: S.DDIMSampler(den, steps=6, silent=True, dtype=torch.float64))
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(S, "WIDE_FUSED", fused)
        smp = make()
        torch.manual_seed(5)
        outs[fused] = smp(x1, **kw)
        if fused:
            loop = next(iter(smp._fused_cache.values()))
            assert isinstance(loop, S._FusedLoopWide) and loop.graphs
            torch.manual_seed(5)
            assert torch.equal(smp(x1, **kw), outs[True])
        else:
            assert not smp._fused_cache
    sc = max(1.0, outs[False].abs().max().item())
    e = max_err(outs[True], outs[False])
    print(case, "ADM, captured vs per-statement fp64 loop: max|d|", e, "scale", sc)
    # measured 1.6e-15 .. 4.0e-15 (the fp32 backbone sees bit-identical inputs; the host table vs the device's fp64 libm)
    assert outs[True].dtype == torch.float64 and e < 1e-10 * sc
    if case == "cfg":  # the default, batched form of the captured step: equal up to the backbone's round-off
        monkeypatch.setenv("AZ_CFG_BATCHED", "1")
        monkeypatch.setattr(S, "WIDE_FUSED", True)
        torch.manual_seed(5)
        xb = make()(x1, **kw)
        print("cfg, 2B-batch captured fp64 loop vs per-statement:", max_err(xb, outs[False]))
        assert max_err(xb, outs[False]) < 1e-3
