r"""UNet backbone + fused sampler parity on the GPU against the reference-generated golden
vectors (G5/G6) and the oracle.  Tolerances (fp32, different summation order in conv / GN):
backbone forward 1e-4 abs on O(1) activations; DDIM-64 trajectory 5e-4 abs on |x0| <= ~4."""

import pytest
import torch

from conftest import ATTN_OPS, DIRECT_OPS, WINO_OPS, max_err
from oracle import nets, sampling, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def build_unet(cfg):
    from azula_amd.nn import UNet

    return UNet(
        cfg["in_channels"], cfg["out_channels"], hid_channels=cfg["hid_channels"], hid_blocks=cfg["hid_blocks"],
        norm=cfg["norm"], groups=cfg["groups"], mod_features=cfg["mod_features"],
    )


@pytest.mark.parametrize("name", ["unet_group", "unet_layer_odd", "unet_rms_nomod"])
def test_unet_forward_matches_reference(golden, name):
    g = golden("g5_" + name)
    cfg = g.meta["cfg"]
    net = build_unet(cfg)
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
    net.load_state_dict(sd)
    net = net.cuda().eval()
    x = g["x"].cuda()
    y = net(x, g["modB"].cuda() if "modB" in g else None)
    err = max_err(y, g["y_modB"])
    print(name, "max|d| vs reference:", err, "scale", g["y_modB"].abs().max().item())
    assert err < 1e-4 * max(1.0, g["y_modB"].abs().max().item())
    if "mod1" in g:
        assert max_err(net(x, g["mod1"].cuda()), g["y_mod1"]) < 1e-4 * max(1.0, g["y_mod1"].abs().max().item())
    # determinism: same launch twice -> bitwise equal
    assert torch.equal(net(x, g["modB"].cuda() if "modB" in g else None), y)


def test_unet_reload_weights_invalidates_plan(golden):
    g = golden("g5_unet_group")
    cfg = g.meta["cfg"]
    net = build_unet(cfg).cuda()
    x, mod = g["x"].cuda(), g["modB"].cuda()
    y0 = net(x, mod)
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
    net.load_state_dict(sd)
    y1 = net(x, mod)
    assert max_err(y1, g["y_modB"]) < 1e-4 * g["y_modB"].abs().max().item()
    assert not torch.equal(y0, y1)


def wrapped_denoiser(g):
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated
    from azula_amd.noise import VPSchedule

    cfg = g.meta["cfg"]
    w = TimeModulated(build_unet(cfg), cfg["mod_features"], name="unet")
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
    w.load_state_dict(sd)
    return KarrasDenoiser(w, VPSchedule()).cuda().eval(), sd, cfg


def test_denoiser_call_matches_reference(golden):
    g = golden("g6_unet_loop")
    den, _, _ = wrapped_denoiser(g)
    q = den(g["x1"].cuda(), torch.tensor(0.5, device="cuda"))
    assert max_err(q.mean, g["mean_t05"]) < 1e-4 * max(1.0, g["mean_t05"].abs().max().item())


def test_ddim64_fused_graph_matches_reference(golden):
    from azula_amd.sample import DDIMSampler

    g = golden("g6_unet_loop")
    den, _, _ = wrapped_denoiser(g)
    smp = DDIMSampler(den, steps=64, silent=True)
    x1 = g["x1"].cuda()
    x0 = smp(x1)
    ent = next(iter(smp._fused_cache.values()))
    assert ent.graph is not None and ent.graph.num_nodes >= len(ent.tape)
    err = max_err(x0, g["ddim64"])
    print("DDIM-64 fused max|d| vs reference:", err, "scale", g["ddim64"].abs().max().item())
    assert err < 5e-4 * max(1.0, g["ddim64"].abs().max().item())
    assert torch.equal(x1.cpu(), g["x1"])  # input not mutated
    assert torch.equal(smp(x1), x0)  # graph replay is deterministic


def test_fused_equals_generic_step_loop(golden):
    """The captured-graph path and the reference-style python loop over step() agree."""
    from azula_amd.sample import DDIMSampler

    g = golden("g6_unet_loop")
    den, _, _ = wrapped_denoiser(g)
    x1 = g["x1"].cuda()
    fused = DDIMSampler(den, steps=8, silent=True)(x1)

    class Loop(DDIMSampler):  # overriding step disables the fused path
        def step(self, x_t, t, s, **kw):
            return super().step(x_t, t, s, **kw)

    generic = Loop(den, steps=8, silent=True)(x1)
    assert max_err(fused, generic) < 1e-5 * max(1.0, fused.abs().max().item())


def test_ddpm8_with_device_rng(golden):
    """DDPM noise comes from torch's device generator, one normal_() per step after the
    backbone, exactly like the reference's randn_like; compare with the oracle fed the same eps."""
    from azula_amd.sample import DDPMSampler

    g = golden("g6_unet_loop")
    den, sd, cfg = wrapped_denoiser(g)
    x1 = g["x1"].cuda()
    torch.manual_seed(123)
    eps = [torch.randn_like(x1).cpu() for _ in range(8)]
    torch.manual_seed(123)
    x0 = DDPMSampler(den, steps=8, silent=True)(x1)
    omean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, cfg, a, c), x, t)  # noqa: E731
    ref = sampling.sample(omean, g["x1"], steps=8, eta=None, eps_list=eps)
    assert max_err(x0, ref) < 5e-4 * max(1.0, ref.abs().max().item())
    # and against the golden DDPM-8 through the generic step() with CPU-recorded noise
    assert g["ddpm8"].shape == x0.shape


def test_unet_forward_through_winograd(golden, monkeypatch):
    """The same reference vectors with every stride-1 3x3 conv forced through the Winograd kernel
    (at these tiny shapes the default policy would pick the direct kernel)."""
    from azula_amd import engine

    monkeypatch.setattr(engine, "WINOGRAD", "2")
    for name in ("unet_group", "unet_layer_odd"):
        g = golden("g5_" + name)
        cfg = g.meta["cfg"]
        net = build_unet(cfg)
        net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
        net = net.cuda().eval()
        y = net(g["x"].cuda(), g["modB"].cuda())
        plan = next(iter(net._plans.values()))
        assert any(name_ in WINO_OPS for _, _, name_ in plan.tape.ops)
        err, sc = max_err(y, g["y_modB"]), g["y_modB"].abs().max().item()
        print(name, "winograd max|d| vs reference:", err, "scale", sc)
        assert err < 2e-4 * max(1.0, sc)


@pytest.mark.parametrize("mode,entry", [("bf16x3", "az_conv2d_x3_f32"), ("f16x2", "az_conv2d_f16x2_f32")])
def test_unet_forward_through_bf16x3(golden, monkeypatch, mode, entry):
    """The same reference vectors with the fp32 convolutions evaluated on the bf16 MFMA as 3 x bf16 pieces / 6 partial
    products (AZ_FP32_MFMA=bf16x3, az_conv2d_x3_f32) or on the f16 MFMA as 2 x f16 pieces / 3 partial products (f16x2,
    az_conv2d_f16x2_f32): same tolerance as the native fp32 path."""
    from azula_amd import engine

    monkeypatch.setattr(engine, "FP32_MFMA", mode)
    monkeypatch.setattr(engine, "X3_MIN_CHANNELS", 4)  # the golden UNets are narrow
    for name in ("unet_group", "unet_layer_odd"):
        g = golden("g5_" + name)
        cfg = g.meta["cfg"]
        net = build_unet(cfg)
        net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
        net = net.cuda().eval()
        y = net(g["x"].cuda(), g["modB"].cuda())
        plan = next(iter(net._plans.values()))
        assert any(name_ == entry for _, _, name_ in plan.tape.ops)
        err, sc = max_err(y, g["y_modB"]), g["y_modB"].abs().max().item()
        print(name, mode, "max|d| vs reference:", err, "scale", sc)
        assert err < 1e-4 * max(1.0, sc)


def test_f16x2_runs_only_on_bounded_inputs(golden, monkeypatch):
    """AZ_FP32_MFMA=f16x2: the kernels whose activation operand has a stated range (|x| < ~1e6) take only inputs whose magnitude does
    not scale with the state -- outputs of normalisations and of convolutions over them (engine.Act.bounded); the layers that read
    the residual / input stream stay on the bf16x3 kernels, whose domain is all of fp32.  So a state of 1e7 (an unstable multistep
    sampler produces one in test_multistep_and_pc_samplers_on_gpu) goes through: finite, equal to the bf16x3 plan's result."""
    from azula_amd import engine

    monkeypatch.setattr(engine, "X3_MIN_CHANNELS", 4)  # the golden UNet is narrow
    monkeypatch.setattr(engine, "WINOGRAD", "2")
    g = golden("g5_unet_group")
    cfg = g.meta["cfg"]
    outs = {}
    for mode in ("bf16x3", "f16x2"):
        monkeypatch.setattr(engine, "FP32_MFMA", mode)
        net = build_unet(cfg)
        net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
        net = net.cuda().eval()
        outs[mode] = [net(g["x"].cuda() * s, g["modB"].cuda()) for s in (1.0, 1.0e7)]
        names = [n for _, _, n in next(iter(net._plans.values())).tape.ops]
        if mode == "f16x2":  # both families on one tape: f16x2 behind the norms, bf16x3 on the stream
            assert any(n.endswith("f16x2_f32") for n in names) and any(n in ("az_conv2d_x3_f32", "az_conv2d_winograd_x3_f32") for n in names), names
    for a, b in zip(outs["f16x2"], outs["bf16x3"]):
        assert torch.isfinite(a).all()
        sc = b.abs().max().item()
        print("f16x2 vs bf16x3 plan:", max_err(a, b), "scale", sc)
        assert max_err(a, b) < 2e-5 * sc


def test_next_samplers_on_gpu(golden):
    """Euler and Ito ride the fused transition kernel (folded coefficients); Heun is a two-evaluation step
    captured as ONE graph (two table rows per step); all against reference-generated vectors (G8)."""
    from azula_amd.sample import EulerSampler, HeunSampler, ItoSampler

    g = golden("g8_unet_next_samplers")
    den, sd, cfg = wrapped_denoiser(g)
    x1 = g["x1"].cuda()
    sc = max(1.0, g["euler16"].abs().max().item())
    smp = EulerSampler(den, steps=16, silent=True)
    x0 = smp(x1)
    assert next(iter(smp._fused_cache.values())).graph is not None
    print("euler16", max_err(x0, g["euler16"]))
    assert max_err(x0, g["euler16"]) < 5e-4 * sc
    smp = HeunSampler(den, steps=8, silent=True)
    x0 = smp(x1)
    loop = next(iter(smp._fused_cache.values()))
    assert loop.graph is not None and loop.n_rows == 16, "Heun must run as a captured two-evaluation graph"
    assert sum(n == "az_step_begin" for _, _, n in loop.tape.ops) == 2
    print("heun8", max_err(x0, g["heun8"]))
    assert max_err(x0, g["heun8"]) < 5e-4 * sc
    assert torch.equal(smp(x1), x0)  # replay determinism

    class GenericHeun(HeunSampler):  # an overridden step() forces the Python-driven loop
        def step(self, x_t, t, s, **kw):
            return super().step(x_t, t, s, **kw)

    xg = GenericHeun(den, steps=8, silent=True)(x1)
    print("heun8 fused vs generic", max_err(x0, xg))
    assert max_err(x0, xg) < 1e-4 * sc
    # Ito with the device RNG: compare with the oracle fed the same noise
    torch.manual_seed(5)
    eps = [torch.randn_like(x1).cpu() for _ in range(16)]
    torch.manual_seed(5)
    x0 = ItoSampler(den, steps=16, silent=True)(x1)
    omean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, cfg, a, c), x, t)  # noqa: E731
    ref = sampling.sample_ito(omean, g["x1"], steps=16, eps_list=eps)
    print("ito16", max_err(x0, ref))
    assert max_err(x0, ref) < 5e-4 * max(1.0, ref.abs().max().item())


def test_multistep_and_pc_samplers_on_gpu(golden):
    """the AB family steps through az_multistep_f32 (one pass per step, host fp64 solves),
    PC through the transition kernel; against reference-generated vectors (G9)."""
    import azula_amd.sample as S

    g = golden("g9_unet_multistep")
    den, sd, cfg = wrapped_denoiser(g)
    x1 = g["x1"].cuda()
    for kind in ("zAB", "vAB", "zEAB", "xEAB", "REAB"):
        want = g[f"{kind}_o3"]
        smp = getattr(S, kind + "Sampler")(den, order=3, steps=g.meta["steps"], silent=True)
        x0 = smp(x1)
        loop = next(iter(smp._fused_cache.values()))
        # ring-slot addresses cycle with period `order`: one graph of 3 steps (+ one for the remainder), replayed
        assert loop.period == 3 and loop.graph is not None, kind
        assert set(loop.graphs) == {3} | ({g.meta["steps"] % 3} if g.meta["steps"] % 3 else set())
        err, sc = max_err(x0, want), max(1.0, want.abs().max().item())
        print(kind, "order 3 max|d| vs reference:", err, "scale", sc)
        assert x0.is_cuda and err < 1e-3 * sc, kind
        assert torch.equal(smp(x1), x0), "the history ring must be reset between calls"
    # other orders against the oracle (order 1 = no history, order 4 with a remainder group, order 2 even split)
    for kind, order, steps in (("zAB", 1, 5), ("xEAB", 4, 10), ("zEAB", 2, 8), ("vAB", 8, 9)):
        smp = getattr(S, kind + "Sampler")(den, order=order, steps=steps, silent=True)
        x0 = smp(x1)
        assert next(iter(smp._fused_cache.values())).graph is not None
        omean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, cfg, a, c), x, t)  # noqa: E731
        ref = sampling.sample_multistep(omean, g["x1"], kind, order=order, steps=steps)
        err, sc = max_err(x0, ref), max(1.0, ref.abs().max().item())
        print(kind, "order", order, "steps", steps, "max|d| vs oracle:", err, "scale", sc)
        assert err < 1e-3 * sc, (kind, order)
    torch.manual_seed(5)
    eps = [torch.randn_like(x1).cpu() for _ in range(g.meta["pc_steps"])]
    torch.manual_seed(5)
    smp = S.PCSampler(den, corrections=1, steps=g.meta["pc_steps"], silent=True)
    x0 = smp(x1)
    loop = next(iter(smp._fused_cache.values()))
    assert loop.graph is not None and loop.n_rows == 2 * g.meta["pc_steps"], "PC must run as a captured graph"
    omean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, cfg, a, c), x, t)  # noqa: E731
    ref = sampling.sample_pc(omean, g["x1"], steps=g.meta["pc_steps"], corrections=1, eps_list=eps)
    print("pc", max_err(x0, ref))
    assert max_err(x0, ref) < 5e-4 * max(1.0, ref.abs().max().item())
    # two corrector moves per step: the noise draws keep the reference's order (two per step, none for the predictor)
    torch.manual_seed(6)
    eps = [torch.randn_like(x1).cpu() for _ in range(2 * 4)]
    torch.manual_seed(6)
    x0 = S.PCSampler(den, corrections=2, delta=0.05, steps=4, silent=True)(x1)
    ref = sampling.sample_pc(omean, g["x1"], steps=4, corrections=2, delta=0.05, eps_list=eps)
    print("pc x2", max_err(x0, ref))
    assert max_err(x0, ref) < 5e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16])
def test_half_precision_module_is_accepted(golden, half):
    """Mirror of the reference's fp16 check (tests/test_nn_unet.py:78-91: q99 < 1e-3, max < 1e-2 between the fp16
    and the fp32 forward), for a module cast with .half() / .bfloat16(): parameters are up-converted once and the
    arithmetic stays fp32, so the only error is the rounding of the weights, inputs and outputs."""
    g = golden("g5_unet_group")
    net = build_unet(g.meta["cfg"])
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    net = net.cuda().eval()
    x, mod = g["x"].cuda(), g["modB"].cuda()
    y32 = net(x, mod)
    net.to(half)
    y16 = net(x.to(half), mod.to(half))
    assert y16.dtype == half and y16.shape == y32.shape
    err = (y32 - y16.float()).abs().flatten()
    scale = y32.abs().max().item()
    tol = 1.0 if half == torch.float16 else 8.0  # bf16 has 3 fewer mantissa bits
    print(half, "q99", torch.quantile(err, 0.99).item(), "max", err.max().item(), "scale", scale)
    assert torch.quantile(err, 0.99) < 1e-3 * tol * max(1.0, scale) and err.max() < 1e-2 * tol * max(1.0, scale)
    # and through a denoiser + fused sampler (KarrasDenoiser casts c_in x_t to the module dtype on the generic path)
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler

    g6 = golden("g6_unet_loop")
    w = TimeModulated(build_unet(g6.meta["cfg"]), g6.meta["cfg"]["mod_features"], name="unet")
    w.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g6.meta["shapes"].items()}, g6.meta["weight_seed"]))
    den = KarrasDenoiser(w, VPSchedule()).cuda().eval()
    x1 = g6["x1"].cuda()
    ref = DDIMSampler(den, steps=8, silent=True)(x1)
    den.backbone.to(half)
    smp = DDIMSampler(den, steps=8, silent=True)
    x0 = smp(x1)
    assert x0.dtype == torch.float32 and next(iter(smp._fused_cache.values())).graph is not None
    q = den(x1, torch.tensor(0.5, device="cuda"))  # generic path: input rounded to the module dtype like the reference
    assert q.mean.dtype == torch.float32 and torch.isfinite(q.mean).all()
    rel = (x0 - ref).abs().max().item() / ref.abs().max().item()
    print(half, "DDIM-8 with half-precision weights vs fp32 weights: rel", rel)
    assert rel < (2e-2 if half == torch.float16 else 1e-1)


@pytest.mark.parametrize("B,H,W", [(1, 8, 8), (2, 9, 5), (3, 1, 1), (1, 2, 3), (5, 16, 1)])
def test_degenerate_spatial_sizes(B, H, W):
    """Four levels on images down to 1 x 1 (every level then sees a single pixel, the stride-2 convs and the
    nearest-upsample / narrow path of azula/nn/unet.py:253-255 run on 1-pixel maps) against the oracle."""
    cfg = dict(in_channels=3, out_channels=3, hid_channels=(8, 16, 32, 64), hid_blocks=(1, 1, 1, 1), norm="group", groups=4,
               mod_features=16)
    net = build_unet(cfg)
    sd = synth.synth_state_dict(synth.shapes_of(net.state_dict()), seed=5)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(B * 100 + H * 10 + W)
    x, mod = torch.randn(B, 3, H, W, generator=g), torch.randn(B, 16, generator=g)
    ref = nets.unet_forward(sd, cfg, x, mod)
    assert max_err(net(x.cuda(), mod.cuda()), ref) < 1e-4 * max(1.0, ref.abs().max().item())
