r"""Parity at BASELINE.json's FULL configuration (configs[1]: ADM-shaped azula UNet, 320.5 M parameters, 4 x 3 x 256 x 256)
through size-independent properties -- the oracle cannot run this size in test time (2 s per step-image on 64 cores):

* the exact-fp32 fused Winograd kernel against the direct implicit-GEMM kernel on every 3x3 layer of the real shapes;
* sample independence: a batch of 4 equals two batches of 2 (different tile counts / split-K choices per launch);
* DDIM(eta = 1) == DDPM on identical noise; the captured step graph == the generic Python step loop;
* replay determinism (bitwise).

Three denoise steps each: the properties do not depend on the number of steps."""

import pytest
import torch

from conftest import max_err

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
STEPS = 3


WINO = ("az_conv2d_winograd_f32", "az_conv2d_winograd_x3_f32", "az_conv2d_winograd_f16x2_f32")  # the Winograd entries (fp32 stream / frequency GEMMs on the bf16 / f16 pipe)


@pytest.fixture(scope="module")
def c2():
    import bench

    cfg = dict(bench.CONFIGS["c2"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    from azula_amd.sample import DDIMSampler

    torch.manual_seed(1)
    x1 = DDIMSampler(den, steps=STEPS, silent=True).init((4, *cfg["shape"]), device="cuda")
    return den, x1


def test_winograd_equals_direct_on_the_real_shapes(c2, monkeypatch):
    from azula_amd import engine

    den, x1 = c2
    t = torch.tensor(0.5, device="cuda")
    net = den.backbone.net
    fast = den(x1, t).mean
    plan = next(iter(net._plans.values()))
    assert sum(n in WINO for _, _, n in plan.tape.ops) >= 40
    monkeypatch.setattr(engine, "WINOGRAD", "0")
    net._plans.clear()
    direct = den(x1, t).mean
    plan = next(iter(net._plans.values()))
    assert not any(n in WINO for _, _, n in plan.tape.ops)
    net._plans.clear()
    scale = direct.abs().max().item()
    print("full-size mean: Winograd vs direct max|d|", max_err(fast, direct), "scale", scale)
    assert max_err(fast, direct) < 5e-6 * max(1.0, scale)  # measured 1.0e-6 on scale 1.25 (MI355X, round 5): bound = 5 x


def test_batch_of_4_equals_two_batches_of_2(c2):
    from azula_amd.sample import DDIMSampler

    den, x1 = c2
    smp = DDIMSampler(den, steps=STEPS, silent=True)
    full = smp(x1)
    assert torch.equal(full, smp(x1)), "graph replay must be deterministic"
    halves = torch.cat([DDIMSampler(den, steps=STEPS, silent=True)(x1[i : i + 2]) for i in (0, 2)])
    scale = full.abs().max().item()
    print("batch 4 vs 2 + 2 max|d|", max_err(full, halves), "scale", scale)
    assert max_err(full, halves) < 1e-6 * max(1.0, scale)  # measured 4.2e-7 on scale 2.5 (other split-K choices at batch 2): bound = 6 x


def test_ddim_eta1_equals_ddpm_and_fused_equals_generic(c2):
    from azula_amd.sample import DDIMSampler, DDPMSampler

    den, x1 = c2
    torch.manual_seed(7)
    a = DDPMSampler(den, steps=STEPS, silent=True)(x1)
    torch.manual_seed(7)
    b = DDIMSampler(den, eta=1.0, steps=STEPS, silent=True)(x1)
    scale = a.abs().max().item()
    assert max_err(a, b) < 1e-5 * max(1.0, scale)

    class Generic(DDIMSampler):  # an overridden step() forces the generic loop (one denoiser call + one kernel per step)
        def step(self, x_t, t, s, **kwargs):
            return super().step(x_t, t, s, **kwargs)

    fused = DDIMSampler(den, steps=STEPS, silent=True)(x1)
    generic = Generic(den, steps=STEPS, silent=True)(x1)
    print("fused vs generic max|d|", max_err(fused, generic), "scale", fused.abs().max().item())
    assert max_err(fused, generic) < 1e-6 * max(1.0, fused.abs().max().item())  # measured 4.8e-7 on scale 2.5: bound = 5 x


def test_dit_b2_full_size_properties():
    """BASELINE configs[2] at its stated batch: DiT-B/2 on 4 x 32 x 32 latents, batch 64, DDIM."""
    import bench
    from azula_amd.sample import DDIMSampler

    cfg = dict(bench.CONFIGS["c3"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    torch.manual_seed(1)
    smp = DDIMSampler(den, steps=STEPS, silent=True)
    x1 = smp.init((cfg["batch"], *cfg["shape"]), device="cuda")
    assert x1.shape[0] == 64
    full = smp(x1)
    assert torch.equal(full, smp(x1))
    halves = torch.cat([DDIMSampler(den, steps=STEPS, silent=True)(x1[i : i + 32]) for i in (0, 32)])
    scale = max(1.0, full.abs().max().item())
    print("DiT-B/2 batch 64 vs 32 + 32 max|d|", max_err(full, halves), "scale", scale)
    assert max_err(full, halves) < 1.5e-6 * scale  # measured 7.2e-7 on scale 2.7: bound = 5.6 x


def test_adm_256_cfg_full_size_properties():
    """BASELINE configs[4] architecture (imagenet_256x256_cond, random init): classifier-free guidance with g = 0 is
    the conditional denoiser itself (azula/guidance/cfg.py:63-65), on the fused two-program graph; batch independence."""
    import bench
    from azula_amd.sample import DDIMSampler

    cfg = dict(bench.CONFIGS["c5cfg"])
    guided = bench.build_denoiser(cfg, torch.device("cuda"))  # CFGDenoiser(AblatedDenoiser)
    plain = guided.denoiser
    torch.manual_seed(1)
    x1 = DDIMSampler(plain, steps=STEPS, silent=True).init((2, *cfg["shape"]), device="cuda")
    lab = torch.tensor([3, 977], device="cuda")
    ref = DDIMSampler(plain, steps=STEPS, silent=True)(x1, label=lab)
    smp = DDIMSampler(guided, steps=STEPS, silent=True)
    g0 = smp(x1, positive={"label": lab}, negative={"label": torch.zeros_like(lab)}, guidance=0.0)
    assert next(iter(smp._fused_cache.values())).graph is not None
    scale = max(1.0, ref.abs().max().item())
    print("ADM-256 CFG(g=0) vs conditional max|d|", max_err(g0, ref), "scale", scale)
    # the guided run evaluates both label sets as ONE 2B batch (other tile / split-K choices than the batch-B plan);
    # ADM's c_out = -100 at t = 1 amplifies those fp32 round-off differences: the ADM trajectory bound applies
    assert max_err(g0, ref) < 9e-4 * scale  # measured 1.8e-4: bound = 5 x
    one = DDIMSampler(plain, steps=STEPS, silent=True)(x1[1:], label=lab[1:])
    # a batch of 1 takes other split-K / tile choices; ADM's c_out = -sigma/alpha = -100 at t = 1 amplifies the fp32
    # round-off differences of the backbone (the same 1e-3 bound as the ADM trajectory tests)
    print("ADM-256 batch 2 vs 1 max|d|", max_err(ref[1:], one))
    assert max_err(ref[1:], one) < 7.5e-4 * scale  # measured 1.5e-4: bound = 5 x


def test_adm_256_cfg_at_baseline_batch_32():
    """BASELINE configs[4] at its stated batch: 32 images, classifier-free guidance = 64 backbone evaluations per step as
    ONE batch-64 plan.  g = 0 must reproduce the conditional denoiser (azula/guidance/cfg.py:63-65), whose batch-32 plan
    makes other tile / split-K choices; replay determinism; sample independence (32 = 16 + 16)."""
    import bench
    from azula_amd.sample import DDIMSampler

    cfg = dict(bench.CONFIGS["c5cfg32"])
    guided = bench.build_denoiser(cfg, torch.device("cuda"))
    plain = guided.denoiser
    B, steps = cfg["batch"], 2
    assert B == 32
    torch.manual_seed(1)
    x1 = DDIMSampler(plain, steps=steps, silent=True).init((B, *cfg["shape"]), device="cuda")
    lab = (torch.arange(B, device="cuda") * 31) % 1000
    ref = DDIMSampler(plain, steps=steps, silent=True)(x1, label=lab)
    smp = DDIMSampler(guided, steps=steps, silent=True)
    kw = dict(positive={"label": lab}, negative={"label": torch.zeros_like(lab)})
    g0 = smp(x1, guidance=0.0, **kw)
    ent = next(iter(smp._fused_cache.values()))
    assert ent.graph is not None and len(ent.fused.programs) == 2
    scale = max(1.0, ref.abs().max().item())
    print("ADM-256 batch 32: CFG(g=0) vs conditional max|d|", max_err(g0, ref), "scale", scale)
    assert max_err(g0, ref) < 1e-3 * scale  # measured 2.1e-4: bound = 4.7 x
    g2 = smp(x1, guidance=2.0, **kw)
    assert torch.equal(g2, smp(x1, guidance=2.0, **kw)) and not torch.equal(g2, g0)
    del smp, ent
    torch.cuda.empty_cache()
    halves = torch.cat([DDIMSampler(plain, steps=steps, silent=True)(x1[i : i + 16], label=lab[i : i + 16]) for i in (0, 16)])
    print("ADM-256 batch 32 vs 16 + 16 max|d|", max_err(ref, halves))
    assert max_err(ref, halves) < 1e-3 * scale  # measured 2.0e-4: bound = 4.9 x


def test_adm_256_ddpm_at_the_c4_shard_of_32():
    """BASELINE configs[3]: ADM 256 x 256 (unconditional), DDPMSampler, the 32-image shard one GPU owns of the 8-way
    split batch of 256.  DDPM == DDIM(eta = 1) on the same device noise; the sharded draw (full-batch noise, this
    rank's slice) equals the slice of the single-device run; replay determinism."""
    import bench
    from azula_amd.sample import DDIMSampler, DDPMSampler

    cfg = dict(bench.CONFIGS["c4"])
    den = bench.build_denoiser(cfg, torch.device("cuda"))
    B, steps = cfg["batch"], 2
    assert B == 32
    torch.manual_seed(1)
    smp = DDPMSampler(den, steps=steps, silent=True)
    x1 = smp.init((B, *cfg["shape"]), device="cuda")
    torch.manual_seed(7)
    a = smp(x1)
    assert next(iter(smp._fused_cache.values())).graph is not None
    torch.manual_seed(7)
    assert torch.equal(a, smp(x1))
    torch.manual_seed(7)
    b = DDIMSampler(den, eta=1.0, steps=steps, silent=True)(x1)
    scale = max(1.0, a.abs().max().item())
    print("ADM-256 DDPM vs DDIM(eta=1), batch 32: max|d|", max_err(a, b), "scale", scale)
    assert max_err(a, b) < 1e-6 * scale  # measured 0.0: the same kernels on the same table values
    # rank 1 of a world of 2 over a global batch of 64: draws 64 x noise per step, keeps rows 32..63
    torch.manual_seed(3)
    big = DDPMSampler(den, steps=steps, silent=True)
    x64 = torch.cat([x1, x1.flip(0)])
    full = big(x64)
    del big
    torch.cuda.empty_cache()
    torch.manual_seed(3)
    smp.shard = (1, 2)
    mine = smp(x64[32:])
    smp.shard = None
    print("ADM-256 shard (rank 1 of 2) vs rows 32..63 of the batch-64 run: max|d|", max_err(mine, full[32:]))
    assert max_err(mine, full[32:]) < 1e-3 * scale  # measured 2.2e-4 (batch 32 vs batch 64 plans): bound = 4.5 x


def test_odd_image_size_through_both_conv_paths(monkeypatch):
    """A mid-size UNet on a 3 x 250 x 190 image (ragged Winograd tiles at every level, odd sizes through the stride-2
    / nearest-upsample / narrow path of azula/nn/unet.py:253-255): Winograd plan == direct plan."""
    from azula_amd import engine
    from azula_amd.nn import UNet

    torch.manual_seed(3)
    net = UNet(3, 3, hid_channels=(64, 128, 256), hid_blocks=(1, 1, 1), norm="group", groups=8, mod_features=64).cuda().eval()
    for blk in net.modules():  # un-zero the AdaZero gates so that every block contributes
        if hasattr(blk, "ada_zero") and isinstance(blk.ada_zero, torch.nn.Sequential):
            blk.ada_zero[-2].weight.data.mul_(100.0)
    x = torch.randn(2, 3, 250, 190, device="cuda")
    mod = torch.randn(64, device="cuda")
    fast = net(x, mod)
    assert any(n in WINO for _, _, n in next(iter(net._plans.values())).tape.ops)
    monkeypatch.setattr(engine, "WINOGRAD", "0")
    net._plans.clear()
    direct = net(x, mod)
    net._plans.clear()
    scale = max(1.0, direct.abs().max().item())
    print("odd-size UNet: Winograd vs direct max|d|", max_err(fast, direct), "scale", scale)
    assert fast.shape == (2, 3, 250, 190) and max_err(fast, direct) < 2.5e-6 * scale  # measured 5.2e-7: bound = 5 x


@pytest.mark.parametrize("H,Cin,Cout", [(256, 256, 256), (64, 512, 512), (16, 1024, 1024)])
@pytest.mark.parametrize("algo", ["winograd", "direct"])
def test_real_layer_shapes_against_torch_cpu(H, Cin, Cout, algo):
    """The layers that carry C2's FLOPs, at their real shapes (one sample), against torch's CPU fp32 convolution."""
    import math

    import torch.nn.functional as F

    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(H + Cin)
    x = torch.randn(1, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    bld = Builder(torch.device("cuda"))
    xa = Act(x.cuda().permute(0, 2, 3, 1).contiguous().reshape(-1), 1, H, H, Cin, Cin, True)
    y = bld.conv(xa, bld.pack_conv(w.cuda(), b.cuda()), Cout, winograd=(algo == "winograd"))
    bld.finish()
    bld.tape.run()
    out = y.buf[: Cout * H * H].view(H, H, Cout).permute(2, 0, 1)[None]
    err = max_err(out, ref)
    print(algo, H, Cin, "max|d|", err, "scale", ref.abs().max().item())
    assert err < 1e-5 * max(1.0, ref.abs().max().item())  # measured: Winograd 4.5e-6, direct 1.24e-5 on scale 6.9 (bound 6.9e-5 = 5.6 x the direct kernel's)


@pytest.mark.parametrize("algo", ["winograd", "direct"])
def test_zero_padding_with_more_than_2_GiB_behind_the_first_sample(algo):
    """Regression (round 2): the loaders mark padded taps with an out-of-range buffer offset.  With > 2 GiB of
    activations behind a workgroup's first sample that offset used to be IN range, so the first samples of a large batch
    read another sample instead of zeros (ADM at 256^2, batch 32: the first 16 images were wrong by 10 % of the scale).
    20 x 512 channels x 256^2 = 2.7 GB: every sample of the batch must equal its own batch-1 evaluation, bitwise."""
    import math

    from azula_amd.engine import Act, Builder

    B, H, Cin, Cout = 20, 256, 512, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B * H * H * Cin, device="cuda", generator=g)
    assert x.numel() * 4 > (1 << 31)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, device="cuda", generator=g)

    def run(xs, nb):
        bld = Builder(torch.device("cuda"))
        y = bld.conv(Act(xs, nb, H, H, Cin, Cin, True), bld.pack_conv(w, b), Cout, winograd=(algo == "winograd"))
        bld.finish()
        bld.tape.run()
        return y.buf[: nb * H * H * Cout].view(nb, H, H, Cout).clone()

    full = run(x, B)
    per = H * H * Cin
    for i in (0, 3, 4, B - 1):  # round 1's clamp broke samples 0..3 of this shape
        one = run(x[i * per : (i + 1) * per].clone(), 1)
        assert torch.equal(full[i : i + 1], one), f"sample {i} of the batch differs from its batch-1 evaluation"
