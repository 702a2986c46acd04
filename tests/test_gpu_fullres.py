r"""Oracle parity at FULL RESOLUTION (VERDICT r05 missing #3): BASELINE.json's networks at their real channel plans AND their real
3 x 256 x 256 resolution, one sample, against the CPU oracle -- the one place where "green" still rested on self-comparison
(tests/test_gpu_fullwidth.py runs the oracle at 64 x 64; tests/test_gpu_fullsize.py compares the build with itself).

What only this size exercises end to end: the 4096-workgroup grids of the 256^2 level, the XCD rectangles of the Winograd
workgroup order, tile blocks that never straddle an image, GroupNorm moments summed over 1024 tile-block partials per image,
and -- at batch 4 -- the non-temporal store gate of the matrix kernels' epilogues (outputs >= 256 MiB).

A C2 forward at batch 1 takes the oracle ~3.5 s on 8 host cores, an ADM-256 forward ~3.4 s; each test
runs the oracle 3 - 4 times.  Tolerances <= 5 x the errors measured on MI355X (printed by the tests).
Reference: azula/nn/unet.py:205-259, azula/plugins/adm/_src/unet.py:605-634, azula/denoise.py:287-324, azula/sample.py:242-261."""

import pytest
import torch

from conftest import ATTN_OPS, DIRECT_OPS, WINO_OPS, max_err
from oracle import nets, sampling

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
RES = 256


@pytest.fixture(scope="module")
def c2_full():
    import bench

    cfg = dict(bench.CONFIGS["c2"])
    assert tuple(cfg["shape"]) == (3, RES, RES)
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    ncfg = dict(cfg["net"])
    torch.set_num_threads(min(64, torch.get_num_threads()))
    oracle_mean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, ncfg, a, c), x, t)  # noqa: E731
    torch.manual_seed(31)
    x1 = torch.randn(1, 3, RES, RES)
    return den, x1, oracle_mean


def test_c2_at_full_resolution_against_the_oracle(c2_full):
    """configs[1]: KarrasDenoiser(TimeModulated(UNet (256, 256, 512, 512, 1024, 1024) x 2 blocks)) on ONE 3 x 256 x 256 sample:
    posterior mean at t = 0.6 and DDIM-2, default policy (x3 Winograd on the 256^2 .. 16^2 levels)."""
    from azula_amd.sample import DDIMSampler

    den, x1, oracle_mean = c2_full
    net = den.backbone.net
    net._plans.clear()
    mean = den(x1.cuda(), torch.tensor(0.6, device="cuda")).mean
    ops = [n for _, _, n in next(iter(net._plans.values())).tape.ops]
    assert sum(ops.count(n) for n in WINO_OPS) >= 40, "the 3 x 3 layers must be on the Winograd kernels"
    ref_mean = oracle_mean(x1, torch.tensor(0.6))
    sc = max(1.0, ref_mean.abs().max().item())
    e1 = max_err(mean, ref_mean)
    smp = DDIMSampler(den, steps=2, silent=True)
    x0 = smp(x1.cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None
    ref_x0 = sampling.sample(oracle_mean, x1, steps=2, eta=0.0)
    e2 = max_err(x0, ref_x0)
    print(f"C2 @{RES}^2, batch 1: mean(t=.6) max|d| {e1:.3e} (scale {sc:.2f}); DDIM-2 max|d| {e2:.3e} (scale {ref_x0.abs().max().item():.2f})")
    net._plans.clear()
    assert e1 < 2.5e-6 * sc  # measured 5.4e-7 on scale 1.00 (MI355X, round 6): bound = 4.7 x
    assert e2 < 2.5e-6 * max(1.0, ref_x0.abs().max().item())  # measured 5.1e-7 on scale 1.31


def test_c2_batch_4_sample_equals_its_batch_1_evaluation(c2_full):
    """At BASELINE's batch 4 the 256^2 level's outputs are 256 MiB: the matrix kernels' epilogues store with the non-temporal
    hint from that size on (conv_shared.h).  Sample 3 of a batch-4 forward == the batch-1 forward of the same sample (itself
    checked against the oracle above) within round-off: other tile blocks, other workgroup rectangles, the hinted stores."""
    den, x1, oracle_mean = c2_full
    net = den.backbone.net
    torch.manual_seed(32)
    xb = torch.randn(4, 3, RES, RES)
    xb[3] = x1[0]
    t = torch.tensor(0.6, device="cuda")
    net._plans.clear()
    one = den(x1.cuda(), t).mean
    net._plans.clear()
    four = den(xb.cuda(), t).mean
    net._plans.clear()
    sc = max(1.0, one.abs().max().item())
    e = max_err(four[3:4], one)
    ref = oracle_mean(x1, torch.tensor(0.6))
    eo = max_err(four[3:4], ref)
    print(f"C2 @{RES}^2: sample 3 of batch 4 vs its batch-1 evaluation max|d| {e:.3e}; vs the oracle {eo:.3e} (scale {sc:.2f})")
    assert e < 1.5e-6 * sc  # measured 3.0e-7: bound = 5 x
    assert eo < 2.5e-6 * sc  # measured 5.2e-7


def test_adm_256_at_full_resolution_against_the_oracle():
    """configs[3] / [4]: ADM imagenet_256x256 (random init) on ONE 3 x 256 x 256 sample: backbone at index 417, posterior mean
    at t = 0.5 (clipped to +-1 in eval mode) and DDIM-2 (c_out = -100 at t = 1: the same amplification as at 64^2,
    tests/test_gpu_fullwidth.py)."""
    import bench
    from azula_amd.plugins import adm
    from azula_amd.sample import DDIMSampler

    cfg = dict(bench.CONFIGS["c5"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    card = dict(adm.load_cards(adm)[cfg["card"]].config)
    sd = {k: v.detach().cpu() for k, v in den.backbone.state_dict().items()}
    sig = sampling.adm_sigmas(card["discrete_schedule"], card["discrete_steps"])
    bb = lambda a, i, y=None: nets.adm_unet_forward(sd, card, a, i, y)  # noqa: E731
    omean = lambda xx, t: sampling.adm_posterior(bb, xx, t, sig)[0]  # noqa: E731
    sched = lambda t: sampling.vp_schedule(t, 1e-2, 1e-2)  # noqa: E731
    torch.set_num_threads(min(64, torch.get_num_threads()))
    torch.manual_seed(33)
    x1 = torch.randn(1, 3, RES, RES)
    out = den.backbone(x1.cuda(), torch.tensor([417], device="cuda"))
    ops = [n for _, _, n in next(iter(den.backbone._plans.values())).tape.ops]
    assert sum(ops.count(n) for n in ATTN_OPS) >= 8
    ref_out = bb(x1, torch.tensor([417]))
    so = max(1.0, ref_out.abs().max().item())
    e0 = max_err(out, ref_out)
    mean = den(x1.cuda(), torch.tensor(0.5, device="cuda")).mean
    e1 = max_err(mean, omean(x1, torch.tensor(0.5)))
    smp = DDIMSampler(den, steps=2, silent=True)
    x0 = smp(x1.cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None
    ref_x0 = sampling.sample(omean, x1, schedule=sched, steps=2, eta=0.0)
    e2 = max_err(x0, ref_x0)
    print(f"ADM-256 @{RES}^2, batch 1: backbone max|d| {e0:.3e} (scale {so:.2f}); mean(t=.5) {e1:.3e}; DDIM-2 {e2:.3e} "
          f"(|x0| <= {ref_x0.abs().max().item():.2f}, c_out = -100 at t = 1)")
    assert e0 < 2e-5  # measured 4.7e-6 on scale 2.85 (at 64^2: 3.0e-6 on scale 2.3)
    assert e1 < 5e-5  # measured 1.3e-5
    assert e2 < 1e-3  # measured 3.6e-4 (at 64^2, DDIM-3: 1.5e-4 .. 1.9e-4)
