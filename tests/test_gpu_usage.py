r"""API usages off the benchmark's beaten path, on the GPU: per-sample times, an fp64 sampler clock, start / stop other than
(1, 0), non-contiguous latents, every schedule x denoiser x sampler family combination through the fused loop, plan-cache
reuse across batch sizes, ADM with per-sample times and a tensor-valued guidance strength -- each against the REFERENCE's
output for the same call (G20, oracle/make_golden.py --only-g20) or, where the device generator's noise enters, against the
oracle fed that noise.  Bounds are <= 5 x the errors measured on MI355X (printed)."""

import pytest
import torch

from conftest import max_err

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _g20(golden):
    from oracle import nets, sampling, synth
    from azula_amd.nn import TimeModulated, UNet

    g = golden("g20_usages")
    cfg = g.meta["unet_cfg"]
    net = TimeModulated(UNet(**{**cfg, "hid_channels": tuple(cfg["hid_channels"]), "hid_blocks": tuple(cfg["hid_blocks"])}),
                        cfg["mod_features"], name="unet")
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["unet_shapes"].items()}, g.meta["unet_weight_seed"])
    net.load_state_dict(sd)
    bb = lambda a, c, **_: nets.time_wrapped_unet(sd, cfg, a, c)  # noqa: E731
    return g, net, bb, sampling


def test_per_sample_times_start_stop_and_layouts(golden):
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler, DDPMSampler

    g, net, bb, sampling = _g20(golden)
    den = KarrasDenoiser(net, VPSchedule()).cuda().eval()
    x = g["x"].cuda()
    # 1. per-sample times
    q = den(x, g["t_per_sample"].cuda())
    e1 = max_err(q.mean, g["mean_per_sample_t"])
    # 2. start / stop other than (1, 0), DDIM against the reference ...
    e2 = max_err(DDIMSampler(den, start=0.8, stop=0.1, steps=5, silent=True)(x), g["ddim5_08_01"])
    # ... DDPM on a NON-CONTIGUOUS latent against the oracle fed the device generator's noise
    xt = x.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    assert not xt.is_contiguous()
    torch.manual_seed(31)
    eps = [torch.randn_like(x).cpu() for _ in range(4)]
    torch.manual_seed(31)
    x0 = DDPMSampler(den, start=0.8, stop=0.1, steps=4, silent=True)(xt)
    om = lambda xx, t: sampling.karras_mean(bb, xx, t)  # noqa: E731
    ref = sampling.sample(om, g["x"], steps=4, eta=None, start=0.8, stop=0.1, eps_list=eps)
    e3 = max_err(x0, ref)
    # 3. same sampler object, two batch sizes (plan cache), then back
    smp = DDIMSampler(den, steps=3, silent=True)
    a = smp(x); b = smp(x[:2]); c = smp(x)  # noqa: E702
    assert torch.equal(a, c)
    e4 = max_err(a[:2], b)
    # 4. fp64 sampler clock on fp32 latents promotes like the reference (values: G11)
    s64 = DDIMSampler(den, steps=5, silent=True, dtype=torch.float64)
    x64 = s64(x)
    assert x64.dtype == torch.float64
    e5 = max_err(x64, sampling.sample(om, g["x"], steps=5, eta=0.0, dtype=torch.float64))
    print(f"per-sample t {e1:.2e}; DDIM start/stop {e2:.2e}; DDPM start/stop non-contiguous {e3:.2e}; batch 3 vs 2 {e4:.2e}; f64 clock {e5:.2e}")
    # measured 1.9e-5 (t = 0.9: c_out amplifies), 2.2e-6, 1.9e-6, 2.9e-6, 3.1e-6 (MI355X, round 4): bounds <= 5 x
    assert e1 < 9e-5 and e2 < 1e-5 and e3 < 1e-5 and e4 < 1.4e-5 and e5 < 1.5e-5


@pytest.mark.parametrize("sname", ["cosine", "rectified"])
@pytest.mark.parametrize("dname", ["karras", "simple"])
def test_schedules_denoisers_and_sampler_families(golden, sname, dname):
    from azula_amd import sample as S
    from azula_amd.denoise import KarrasDenoiser, SimpleDenoiser
    from azula_amd.noise import CosineSchedule, RectifiedSchedule

    g, net, bb, sampling = _g20(golden)
    sch, sora = {"cosine": (CosineSchedule(), sampling.cosine_schedule), "rectified": (RectifiedSchedule(), sampling.rectified_schedule)}[sname]
    D, dora = {"karras": (KarrasDenoiser, sampling.karras_mean), "simple": (SimpleDenoiser, sampling.simple_mean)}[dname]
    d = D(net, sch).cuda().eval()
    key = f"{sname}_{dname}"
    x1 = g[key + "_x1"].cuda()
    errs = {}
    for tag, smp in (("ddim", S.DDIMSampler(d, steps=4, silent=True)), ("euler", S.EulerSampler(d, steps=4, silent=True)),
                     ("heun", S.HeunSampler(d, steps=4, silent=True)), ("zeab", S.zEABSampler(d, order=2, steps=4, silent=True))):
        out = smp(x1)
        assert smp._fused_cache, (key, tag, "the fused (captured) loop must take these")
        errs[tag] = max_err(out, g[f"{key}_{tag}"]) / max(1.0, g[f"{key}_{tag}"].abs().max().item())
    torch.manual_seed(32)
    eps = [torch.randn_like(x1).cpu() for _ in range(4)]
    torch.manual_seed(32)
    pc = S.PCSampler(d, corrections=1, steps=4, silent=True)(x1)
    om = lambda xx, t: dora(bb, xx, t, schedule=sora)  # noqa: E731
    ref = sampling.sample_pc(om, g[key + "_x1"], schedule=sora, steps=4, corrections=1, eps_list=eps)
    errs["pc"] = max_err(pc, ref) / max(1.0, ref.abs().max().item())
    print(key, {k: f"{v:.2e}" for k, v in errs.items()})
    assert all(v < 3e-5 for v in errs.values()), errs  # measured 7.6e-7 .. 1.3e-5 (Heun: two evaluations per step)


def test_adm_per_sample_times_and_tensor_guidance(golden):
    from oracle import synth
    from azula_amd.guidance import CFGDenoiser
    from azula_amd.plugins import adm
    from azula_amd.sample import DDIMSampler

    g = golden("g20_usages")
    acfg = dict(g.meta["adm_cfg"])
    ad = adm.make_model(**acfg)
    ad.backbone.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["adm_shapes"].items()}, g.meta["adm_weight_seed"]))
    ad = ad.cuda().eval()
    xa, lab = g["adm_x"].cuda(), g["adm_label"].cuda()
    q = ad(xa, g["adm_t"].cuda(), label=lab)
    e1, e2 = max_err(q.mean, g["adm_mean"]), max_err(q.var, g["adm_var"]) / max(1.0, g["adm_var"].abs().max().item())
    o = DDIMSampler(CFGDenoiser(ad), steps=3, silent=True)(xa, positive={"label": lab}, negative={"label": torch.zeros_like(lab)},
                                                           guidance=torch.tensor(1.5))
    e3 = max_err(o, g["adm_cfg_ddim3"])
    print(f"ADM per-sample t: mean {e1:.2e}, var {e2:.2e}; CFG(tensor guidance) DDIM-3 {e3:.2e}")
    # measured 5.2e-5 (c_out = -41 at t = 0.9 amplifies the backbone's 1.3e-6), 1.4e-6, 2.9e-4: bounds <= 5 x
    assert e1 < 2.5e-4 and e2 < 1e-5 and e3 < 1.5e-3


# ------------------------------------------------------------------------------------------------------------------
# configs[0]'s shape on the GPU: a USER-DEFINED nn.Module backbone (the reference tests' 2-layer MLP) on (64, 5) CUDA
# latents.  The backbone is opaque to the engine, so Sampler.__call__ runs the generic loop: torch evaluates the module,
# every step's elementwise work goes through az_transition_f32 on 2-D latents.  Checked against G4 (reference outputs).
def test_user_defined_mlp_backbone_on_cuda_latents(golden):
    from conftest import max_err
    from oracle import sampling, synth
    from azula_amd import _lib
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler, DDPMSampler

    class ToyMLP(torch.nn.Module):  # reference tests/test_sample.py:28-53
        def __init__(self, features=5):
            super().__init__()
            self.l1 = torch.nn.Linear(features, 64)
            self.l2 = torch.nn.Linear(64, features)

        def forward(self, x_t, t, label=None):
            f = torch.exp(torch.log(torch.tensor(1e-4)) * torch.linspace(0, 1, 32, dtype=x_t.dtype)).to(x_t.device)
            e = torch.cat((torch.sin(t.unsqueeze(-1) * f), torch.cos(t.unsqueeze(-1) * f)), dim=-1)
            return self.l2(torch.relu(self.l1(x_t) + e))

    g = golden("g4_toy_loop")
    net = ToyMLP()
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    import copy

    den = KarrasDenoiser(copy.deepcopy(net), VPSchedule()).eval().cuda()
    calls = []
    real = _lib.call
    try:
        _lib.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        x1 = g["x1"].cuda()
        smp = DDIMSampler(den, steps=64, silent=True)
        x0 = smp(x1)
        assert not smp._fused_cache, "an opaque nn.Module must take the generic loop"
        assert calls.count("az_transition_f32") == 64, "one az_transition_f32 per step on the 2-D latents"
        e = max_err(x0, g["ddim64"])
        print("MLP backbone on CUDA (64, 5), DDIM-64 vs the reference's output:", e)
        assert x0.is_cuda and x0.shape == (64, 5) and e < 2e-5  # CPU path: 1e-5; torch's GPU matmul / sin / cos differ in the last ulp

        torch.manual_seed(5)
        eps = [torch.randn_like(x1).cpu() for _ in range(64)]
        torch.manual_seed(5)
        calls.clear()
        x0 = DDPMSampler(den, steps=64, silent=True)(x1)
        assert calls.count("az_transition_f32") == 64
    finally:
        _lib.call = real
    omean = lambda x, t: sampling.karras_mean(lambda a, c: net(a, c), x, t)  # noqa: E731 -- the oracle's Karras mean around the CPU module
    ref = sampling.sample(omean, g["x1"], steps=64, eta=None, eps_list=eps)
    e = max_err(x0, ref)
    print("MLP backbone on CUDA (64, 5), DDPM-64 with the device generator's noise vs the oracle loop:", e)
    assert e < 5e-5
