r"""Oracle parity at the FULL CHANNEL WIDTHS of BASELINE.json's networks, end to end.

The golden fixtures (G5/G6) use channel widths <= 64; the full-size tests (test_gpu_fullsize.py) are self-comparisons.
This file closes the gap between them: the real channel plans -- azula UNet (256, 256, 512, 512, 1024, 1024) x 2 blocks
(configs[1]) and ADM imagenet_256x256 (256 x (1, 1, 2, 2, 4, 4), attention at 1/8, 1/16, 1/32; configs[3] / [4]) -- on a
64 x 64 image, ONE sample, where the CPU oracle needs seconds.  Every layer then runs at its real K (up to 9 x 2048 in the
merge convolutions): the Winograd kernel with K chunks from both skip sources, direct + split-K at K = 9 x 1024 on the
4 x 4 / 2 x 2 maps, GroupNorm over 1024 channels, ADM attention with 16 heads.  With ``AZ_WINOGRAD=2`` semantics (forced
by monkeypatch) every stride-1 3 x 3 layer goes through the Winograd kernel with split-K, down to the 2 x 2 maps.

Tolerances are <= 5 x the errors measured on MI355X (printed by the tests)."""

import pytest
import torch

from conftest import ATTN_OPS, DIRECT_OPS, WINO_OPS, max_err
from oracle import nets, sampling

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
RES = 64


@pytest.fixture(scope="module")
def c2_net():
    import bench

    cfg = dict(bench.CONFIGS["c2"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    ncfg = dict(cfg["net"])
    torch.set_num_threads(min(64, torch.get_num_threads()))
    oracle_mean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, ncfg, a, c), x, t)  # noqa: E731
    torch.manual_seed(11)
    x1 = torch.randn(1, 3, RES, RES)
    ref_mean = oracle_mean(x1, torch.tensor(0.6))
    ref_x0 = sampling.sample(oracle_mean, x1, steps=3, eta=0.0)
    return den, x1, ref_mean, ref_x0


@pytest.mark.parametrize("policy", ["1", "2", "0"])
def test_c2_channel_plan_against_the_oracle(c2_net, policy, monkeypatch):
    from azula_amd import engine
    from azula_amd.sample import DDIMSampler

    den, x1, ref_mean, ref_x0 = c2_net
    monkeypatch.setattr(engine, "WINOGRAD", policy)
    net = den.backbone.net
    net._plans.clear()
    mean = den(x1.cuda(), torch.tensor(0.6, device="cuda")).mean
    ops = [n for _, _, n in next(iter(net._plans.values())).tape.ops]
    nw, nd = sum(ops.count(n) for n in WINO_OPS), sum(ops.count(n) for n in DIRECT_OPS)  # (whatever the AZ_FP32_MFMA mode)
    if policy == "2":
        assert nw >= 50 and nd <= 8, (nw, nd)  # only the stride-2 convolutions and the stem / head stay direct
    elif policy == "0":
        assert nw == 0
    else:
        assert nw >= 20 and nd >= 20, (nw, nd)  # 64^2 .. 16^2 Winograd; 8^2 .. 2^2 direct + split-K at K = 9 x 1024
    sc = max(1.0, ref_mean.abs().max().item())
    e1 = max_err(mean, ref_mean)
    x0 = DDIMSampler(den, steps=3, silent=True)(x1.cuda())
    e2 = max_err(x0, ref_x0)
    print(f"C2 widths @{RES}^2 policy {policy}: {nw} winograd / {nd} direct convs; mean max|d| {e1:.3e} (scale {sc:.2f}); "
          f"DDIM-3 max|d| {e2:.3e} (scale {ref_x0.abs().max().item():.2f})")
    net._plans.clear()
    assert e1 < 2e-6 * sc  # measured 3.3e-7 .. 4.0e-7
    assert e2 < 2e-6 * max(1.0, ref_x0.abs().max().item())  # measured 3.0e-7 .. 3.6e-7


def test_c2_channel_plan_ddim64_full_length(c2_net):
    """BASELINE configs[1]'s sampler at its real length -- DDIMSampler(steps=64) -- at configs[1]'s full channel plan, one
    64 x 64 sample, against the oracle's 64-step loop on the host (the full-width check was DDIM-3)."""
    from azula_amd.sample import DDIMSampler

    den, x1, _, _ = c2_net
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    import bench

    ncfg = dict(bench.CONFIGS["c2"]["net"])
    oracle_mean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, ncfg, a, c), x, t)  # noqa: E731
    den.backbone.net._plans.clear()
    smp = DDIMSampler(den, steps=64, silent=True)
    x0 = smp(x1.cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None
    ref = sampling.sample(oracle_mean, x1, steps=64, eta=0.0)
    sc = max(1.0, ref.abs().max().item())
    e = max_err(x0, ref)
    print(f"C2 widths @{RES}^2 DDIM-64 max|d| {e:.3e} (scale {sc:.2f})")
    den.backbone.net._plans.clear()
    assert e < 3.4e-6 * sc  # measured 2.6e-6 on scale 3.88 = 6.8e-7 relative (MI355X, round 4): bound = 5 x


def test_c2_channel_plan_under_the_f4_policy(c2_net, monkeypatch):
    """The opt-in inexact F(4x4,3x3) policy (AZ_WINOGRAD=4: the fp32 kernel of round 1 on every layer with >= 16 tiles) on
    configs[1]'s channel plan: what the 6 x 6 transforms cost in accuracy END TO END (posterior mean, DDIM-3, DDIM-64) -- the
    accuracy half of the round-6 gate for F(4x4) on the split operands (profiles/r06_wx3_f4_gate.txt)."""
    from azula_amd import engine
    from azula_amd.sample import DDIMSampler

    den, x1, ref_mean, ref_x0 = c2_net
    monkeypatch.setattr(engine, "WINOGRAD", "4")
    monkeypatch.setattr(engine, "WINOGRAD4_MIN_TILES", 16)
    net = den.backbone.net
    net._plans.clear()
    mean = den(x1.cuda(), torch.tensor(0.6, device="cuda")).mean
    ops = [n for _, _, n in next(iter(net._plans.values())).tape.ops]
    n4 = ops.count("az_conv2d_winograd4_f32")
    assert n4 >= 25, n4  # the 64^2, 32^2 and 16^2 levels
    e1 = max_err(mean, ref_mean)
    x0 = DDIMSampler(den, steps=3, silent=True)(x1.cuda())
    e2 = max_err(x0, ref_x0)
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    import bench

    ncfg = dict(bench.CONFIGS["c2"]["net"])
    oracle_mean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, ncfg, a, c), x, t)  # noqa: E731
    x64 = DDIMSampler(den, steps=64, silent=True)(x1.cuda())
    ref64 = sampling.sample(oracle_mean, x1, steps=64, eta=0.0)
    e3 = max_err(x64, ref64)
    print(f"C2 widths @{RES}^2 policy 4 (F(4x4) fp32 on {n4} layers): mean max|d| {e1:.3e} (scale {ref_mean.abs().max().item():.2f}); "
          f"DDIM-3 {e2:.3e} (scale {ref_x0.abs().max().item():.2f}); DDIM-64 {e3:.3e} (scale {ref64.abs().max().item():.2f})")
    net._plans.clear()
    # measured (MI355X, round 6): mean 3.8e-6 (F(2x2): 3.3e-7), DDIM-3 3.6e-6 (3.6e-7), DDIM-64 2.6e-6 on scale 3.88 (2.6e-6: the long
    # trajectory's error is not the convolutions').  Bounds = 5 x.
    assert e1 < 2e-5 and e2 < 2e-5 * max(1.0, ref_x0.abs().max().item()) and e3 < 3.4e-6 * max(1.0, ref64.abs().max().item())


@pytest.fixture(scope="module")
def adm_net():
    import bench
    from azula_amd.plugins import adm

    cfg = dict(bench.CONFIGS["c5"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    card = dict(adm.load_cards(adm)[cfg["card"]].config)
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    sig = sampling.adm_sigmas(card["discrete_schedule"], card["discrete_steps"])
    bb = lambda a, i, y=None: nets.adm_unet_forward(sd, card, a, i, y)  # noqa: E731
    omean = lambda xx, t: sampling.adm_posterior(bb, xx, t, sig)[0]  # noqa: E731
    sched = lambda t: sampling.vp_schedule(t, 1e-2, 1e-2)  # noqa: E731
    torch.manual_seed(12)
    x1 = torch.randn(1, 3, RES, RES)
    ref_out = bb(x1, torch.tensor([417]))
    ref_mean = omean(x1, torch.tensor(0.5))
    ref_x0 = sampling.sample(omean, x1, schedule=sched, steps=3, eta=0.0)
    return den, x1, ref_out, ref_mean, ref_x0, omean, sched


def test_adm_256_widths_against_the_oracle(adm_net):
    from azula_amd.sample import DDIMSampler

    den, x1, ref_out, ref_mean, ref_x0, _, _ = adm_net
    out = den.backbone(x1.cuda(), torch.tensor([417], device="cuda"))
    ops = [n for _, _, n in next(iter(den.backbone._plans.values())).tape.ops]
    assert sum(ops.count(n) for n in ATTN_OPS) >= 8, "the attention blocks at 1/8, 1/16, 1/32 must be on the tape"
    so = max(1.0, ref_out.abs().max().item())
    e0 = max_err(out, ref_out)
    mean = den(x1.cuda(), torch.tensor(0.5, device="cuda")).mean
    e1 = max_err(mean, ref_mean)
    smp = DDIMSampler(den, steps=3, silent=True)
    x0 = smp(x1.cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None
    e2 = max_err(x0, ref_x0)
    print(f"ADM-256 widths @{RES}^2: backbone max|d| {e0:.3e} (scale {so:.2f}); mean(t=.5) {e1:.3e}; DDIM-3 {e2:.3e} "
          f"(|x0| <= {ref_x0.abs().max().item():.2f}, c_out = -100 at t = 1)")
    assert e0 < 1.5e-5  # measured 3.0e-6 on scale 2.3
    assert e1 < 4e-5  # measured 7.9e-6
    assert e2 < 7.5e-4  # measured 1.5e-4 .. 1.9e-4


def test_adm_256_widths_ddpm_against_the_oracle(adm_net):
    """configs[3]'s sampler (DDPM) at configs[3]'s widths: the oracle is fed the device generator's noise."""
    from azula_amd.sample import DDPMSampler

    den, x1, _, _, _, omean, sched = adm_net
    torch.manual_seed(21)
    eps = [torch.randn(1, 3, RES, RES, device="cuda").cpu() for _ in range(3)]
    torch.manual_seed(21)
    x0 = DDPMSampler(den, steps=3, silent=True)(x1.cuda())
    ref = sampling.sample(omean, x1, schedule=sched, steps=3, eta=None, eps_list=eps)
    e = max_err(x0, ref)
    print(f"ADM-256 widths DDPM-3 max|d| {e:.3e} (scale {ref.abs().max().item():.2f})")
    assert e < 1.5e-4  # measured 3.0e-5


# ------------------------------------------------------------------------------------------------------------------
# configs[2]: DiT-B/2 at its REAL width (768 x 12 blocks, 12 heads, patch 2, 4 x 32 x 32 latents -> 256 tokens) and
# JiT-B/16 (768 x 12, 256 + 32 tokens), against the CPU oracle; the golden fixtures G5 / G6 / G10 are hid-64 networks.
def k16_modes():
    return [None, "1"]  # the host's own K-tile choice / the 16-channel K tile forced on every direct-kernel launch


@pytest.fixture(scope="module")
def dit_b2():
    import bench

    cfg = dict(bench.CONFIGS["c3"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    ncfg = dict(cfg["net"])
    torch.set_num_threads(min(64, torch.get_num_threads()))
    oracle_mean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_vit(sd, ncfg, a, c), x, t)  # noqa: E731
    torch.manual_seed(13)
    x1 = torch.randn(2, *cfg["shape"])
    ref_mean = oracle_mean(x1, torch.tensor(0.6))
    ref_x0 = sampling.sample(oracle_mean, x1, steps=3, eta=0.0)
    return den, x1, ref_mean, ref_x0


@pytest.mark.parametrize("k16", k16_modes())
def test_dit_b2_full_width_against_the_oracle(dit_b2, k16, monkeypatch):
    from azula_amd.sample import DDIMSampler

    den, x1, ref_mean, ref_x0 = dit_b2
    if k16 is None:
        monkeypatch.delenv("AZ_IGEMM_K16", raising=False)
    else:
        monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        monkeypatch.setenv("AZ_IGEMM_K16", k16)
    net = den.backbone.net
    net._plans.clear()
    mean = den(x1.cuda(), torch.tensor(0.6, device="cuda")).mean
    ops = [n for _, _, n in next(iter(net._plans.values())).tape.ops]
    assert sum(ops.count(n) for n in ATTN_OPS) == 12 and sum(ops.count(n) for n in DIRECT_OPS) >= 4 * 12
    sc = max(1.0, ref_mean.abs().max().item())
    e1 = max_err(mean, ref_mean)
    smp = DDIMSampler(den, steps=3, silent=True)
    x0 = smp(x1.cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None
    e2 = max_err(x0, ref_x0)
    print(f"DiT-B/2 full width, K16={k16}: mean max|d| {e1:.3e} (scale {sc:.2f}); DDIM-3 max|d| {e2:.3e} "
          f"(scale {ref_x0.abs().max().item():.2f})")
    net._plans.clear()
    assert e1 < 2e-6 * sc  # measured 7.2e-7 on scale 1.88: bound = 5 x
    assert e2 < 1.5e-6 * max(1.0, ref_x0.abs().max().item())  # measured 8.0e-7 on scale 2.65: bound = 5 x


@pytest.mark.parametrize("k16", k16_modes())
def test_jit_b16_full_width_against_the_oracle(k16, monkeypatch):
    import bench

    if k16 is None:
        monkeypatch.delenv("AZ_IGEMM_K16", raising=False)
    else:
        monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        monkeypatch.setenv("AZ_IGEMM_K16", k16)
    cfg = dict(bench.CONFIGS["c6"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    torch.set_num_threads(min(64, torch.get_num_threads()))
    torch.manual_seed(14)
    x = torch.randn(1, 3, 256, 256)
    t, y = torch.tensor([0.37]), torch.tensor([207])
    ref = nets.jit_forward(sd, {"model": cfg["model"], "input_size": 256}, x, t, y)
    out = den.backbone(x.cuda(), t.cuda(), y.cuda())
    sc = max(1.0, ref.abs().max().item())
    e = max_err(out, ref)
    print(f"JiT-B/16 full width, K16={k16}: backbone max|d| {e:.3e} (scale {sc:.2f})")
    assert e < 3.3e-6 * sc  # measured 3.0e-6 on scale 4.57 = 6.5e-7 relative: bound = 5 x
    bb = lambda a, c, lab: nets.jit_forward(sd, {"model": cfg["model"], "input_size": 256}, a, c, lab)  # noqa: E731
    ref_mean = sampling.jit_mean(bb, x, torch.tensor(0.4), y)
    mean = den(x.cuda(), torch.tensor(0.4, device="cuda"), label=y.cuda()).mean
    e = max_err(mean, ref_mean)
    print(f"JiT-B/16 full width, K16={k16}: posterior mean max|d| {e:.3e} (scale {ref_mean.abs().max().item():.2f})")
    assert e < 3.3e-6 * max(1.0, ref_mean.abs().max().item())  # measured 2.9e-6 on scale 4.56: bound = 5 x
