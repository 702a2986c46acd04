r"""``Sampler(dtype=torch.float64)`` and fp64 latents on the GPU (G11, reference-generated): the reference's type promotion
makes the whole elementwise path of such a sampler fp64 while the backbone stays fp32 (azula/sample.py:69-94,
azula/denoise.py:306-322).  Here: ``az_scale_f64_to_f32`` / ``az_axpby_f64`` / ``az_transition_f64`` around the fp32 backbone
plans; the output dtype is fp64 like the reference's.  Tolerance: the fp32 backbone's round-off (1e-5 of the scale)."""

import pytest
import torch

from conftest import max_err
from oracle import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def unet_denoiser(g):
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated, UNet
    from azula_amd.noise import VPSchedule

    cfg = g.meta["unet_cfg"]
    net = UNet(cfg["in_channels"], cfg["out_channels"], hid_channels=cfg["hid_channels"], hid_blocks=cfg["hid_blocks"],
               norm=cfg["norm"], groups=cfg["groups"], mod_features=cfg["mod_features"])
    w = TimeModulated(net, cfg["mod_features"], name="unet")
    w.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["unet_shapes"].items()}, g.meta["unet_weight_seed"]))
    return KarrasDenoiser(w, VPSchedule()).cuda().eval()


def test_fp64_time_grid_matches_reference(golden):
    from azula_amd.sample import DDIMSampler, EulerSampler, HeunSampler, zABSampler

    g = golden("g11_sampler_dtype")
    den = unet_denoiser(g)
    x1 = g["unet_x1"].cuda()
    x0 = DDIMSampler(den, steps=8, silent=True, dtype=torch.float64)(x1)
    sc = max(1.0, g["unet_ddim8"].abs().max().item())
    print("DDIM-8, fp64 grid: max|d|", max_err(x0, g["unet_ddim8"]), "scale", sc)
    assert x0.dtype == torch.float64 and max_err(x0, g["unet_ddim8"]) < 2e-5 * sc
    x32 = DDIMSampler(den, steps=8, silent=True)(x1)
    assert x32.dtype == torch.float32 and max_err(x32, x0) < 1e-4 * sc  # the fp32 sampler agrees to fp32 round-off
    xh = HeunSampler(den, steps=4, silent=True, dtype=torch.float64)(x1)
    print("Heun-4, fp64 grid: max|d|", max_err(xh, g["unet_heun4"]))
    assert xh.dtype == torch.float64 and max_err(xh, g["unet_heun4"]) < 2e-5 * max(1.0, g["unet_heun4"].abs().max().item())
    xe = EulerSampler(den, steps=6, silent=True, dtype=torch.float64)(x1.double())
    print("Euler-6, fp64 latents: max|d|", max_err(xe, g["unet_euler6_x64"]))
    assert xe.dtype == torch.float64 and max_err(xe, g["unet_euler6_x64"]) < 2e-5 * max(1.0, g["unet_euler6_x64"].abs().max().item())
    # the multistep family in fp64 (az_axpby_f64 from an fp64 coefficient table) against its fp32 kernel form
    from azula_amd.sample import xEABSampler

    for cls, order in ((zABSampler, 2), (xEABSampler, 3)):
        xa = cls(den, order=order, steps=6, silent=True, dtype=torch.float64)(x1)
        x32 = cls(den, order=order, steps=6, silent=True)(x1)
        assert xa.dtype == torch.float64 and x32.dtype == torch.float32
        print(cls.__name__, "fp64 grid vs fp32 sampler: max|d|", max_err(xa, x32))
        assert max_err(xa, x32) < 1e-4 * sc  # two precisions of the same integrator around the same fp32 network


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.float64])
def test_fp64_clock_runs_as_a_captured_loop(golden, in_dtype, monkeypatch):
    """``DDIMSampler / DDPMSampler(dtype=float64)``: one hipGraph per step (az_step_row_f64 + az_scale_f64_to_f32 + backbone +
    az_axpby_f64 + az_transition_f64) instead of the per-statement loop -- equal to that loop to fp64 round-off (same kernels,
    the same generator draws: fp32 in the first step of an fp32 input, fp64 afterwards) and within the fp32 backbone's
    round-off of the reference's output (G11)."""
    from azula_amd import sample as S

    g = golden("g11_sampler_dtype")
    den = unet_denoiser(g)
    x1 = g["unet_x1"].cuda().to(in_dtype)
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(S, "WIDE_FUSED", fused)
        smp = S.DDIMSampler(den, steps=8, silent=True, dtype=torch.float64)
        ddim = smp(x1)
        if fused:
            loop = next(iter(smp._fused_cache.values()))
            assert isinstance(loop, S._FusedLoopWide) and loop.graphs[1].num_nodes >= len(loop.tape)
            assert torch.equal(smp(x1), ddim)  # replay
        else:
            assert not smp._fused_cache
        torch.manual_seed(77)
        ddpm = S.DDPMSampler(den, steps=6, silent=True, dtype=torch.float64)(x1)
        torch.manual_seed(77)
        eta = S.DDIMSampler(den, steps=6, eta=0.5, silent=True, dtype=torch.float64)(x1)
        outs[fused] = (ddim, ddpm, eta)
    for a, b, what in zip(outs[True], outs[False], ("DDIM-8", "DDPM-6", "DDIM-6 eta 0.5")):
        # (not bit-equal: the per-statement loop evaluates the schedule with the DEVICE's fp64 libm, the captured loop reads the
        # host table -- the reference's CPU values; measured 1.8e-15)
        assert a.dtype == torch.float64 and max_err(a, b) < 1e-12, (what, max_err(a, b))
    if in_dtype == torch.float32:
        sc = max(1.0, g["unet_ddim8"].abs().max().item())
        e = max_err(outs[True][0], g["unet_ddim8"])
        print("captured fp64 loop, DDIM-8 vs the reference:", e, "scale", sc)
        assert e < 2e-5 * sc


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("which", ["euler", "heun", "ito", "pc"])
def test_fp64_clock_captured_loop_of_the_other_samplers(golden, which, in_dtype, monkeypatch):
    """Euler / Ito (one evaluation per step), Heun (two evaluations, the first posterior mean kept, the spare-coefficient axpby in
    fp64) and PC (corrections + 1 evaluations, one noise buffer per corrector move) with ``dtype=float64``: the captured fp64 loop
    against the per-statement fp64 loop -- the same integrator with the same generator draws (the first randn_like(x_t) of an
    fp32 input is fp32, every later one fp64; Ito's randn_like(x_s) is fp64 from the start, azula/sample.py:427-429).  Euler /
    Heun fold their coefficients in the captured form, so equality is to fp64 round-off, not to the bit."""
    from azula_amd import sample as S

    g = golden("g11_sampler_dtype")
    den = unet_denoiser(g)
    x1 = g["unet_x1"].cuda().to(in_dtype)
    make = {
        "euler": lambda: S.EulerSampler(den, steps=6, silent=True, dtype=torch.float64),
        "heun": lambda: S.HeunSampler(den, steps=4, silent=True, dtype=torch.float64),
        "ito": lambda: S.ItoSampler(den, steps=5, eta=0.7, silent=True, dtype=torch.float64),
        "pc": lambda: S.PCSampler(den, steps=3, corrections=2, silent=True, dtype=torch.float64),
    }[which]
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(S, "WIDE_FUSED", fused)
        smp = make()
        torch.manual_seed(31)
        outs[fused] = smp(x1)
        if fused:
            loop = next(iter(smp._fused_cache.values()))
            assert isinstance(loop, S._FusedLoopWide) and loop.graphs[1].num_nodes >= len(loop.tape)
            torch.manual_seed(31)
            assert torch.equal(smp(x1), outs[True])  # replay of the captured graph
        else:
            assert not smp._fused_cache
    sc = max(1.0, outs[False].abs().max().item())
    e = max_err(outs[True], outs[False])
    print(which, in_dtype, "captured vs per-statement fp64 loop: max|d|", e, "scale", sc)
    assert outs[True].dtype == torch.float64 and e < 1e-10 * sc
    if in_dtype == torch.float32 and which == "heun":
        assert max_err(outs[True], g["unet_heun4"]) < 2e-5 * max(1.0, g["unet_heun4"].abs().max().item())
    if in_dtype == torch.float64 and which == "euler":
        assert max_err(outs[True], g["unet_euler6_x64"]) < 2e-5 * max(1.0, g["unet_euler6_x64"].abs().max().item())


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("which", ["zab2", "xeab3", "vab1", "reab4"])
def test_fp64_clock_captured_loop_of_the_multistep_family(golden, which, in_dtype, monkeypatch):
    """The Adams-Bashforth family with ``dtype=float64``: `order` consecutive steps per captured graph (the history ring's
    addresses cycle), every statement an az_axpby_f64 reading its weight from words 36 .. 47 of the current fp64 row -- against
    the per-statement fp64 loop (``_MultistepSampler._call_wide``), 7 steps (a remainder graph for orders 2 - 4)."""
    from azula_amd import sample as S

    g = golden("g11_sampler_dtype")
    den = unet_denoiser(g)
    x1 = g["unet_x1"].cuda().to(in_dtype)
    cls, order = {"zab2": (S.zABSampler, 2), "xeab3": (S.xEABSampler, 3), "vab1": (S.vABSampler, 1), "reab4": (S.REABSampler, 4)}[which]
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(S, "WIDE_FUSED", fused)
        smp = cls(den, order=order, steps=7, silent=True, dtype=torch.float64)
        outs[fused] = smp(x1)
        if fused:
            loop = next(iter(smp._fused_cache.values()))
            assert isinstance(loop, S._FusedLoopWide) and loop.period == order and loop.graphs
            assert torch.equal(smp(x1), outs[True])  # replay of the captured graphs (ring zeroed per call)
        else:
            assert not smp._fused_cache
    sc = max(1.0, outs[False].abs().max().item())
    e = max_err(outs[True], outs[False])
    print(which, in_dtype, "captured vs per-statement fp64 loop: max|d|", e, "scale", sc)
    assert outs[True].dtype == torch.float64 and e < 1e-10 * sc


@pytest.mark.parametrize("fused", [True, False])
def test_ito_with_an_fp64_clock_draws_its_noise_in_fp64(golden, fused, monkeypatch):
    """ItoSampler's noise is randn_like(x_s) (azula/sample.py:427-429): with fp64 schedule scalars x_s is fp64 before the draw, also
    in the first step of an fp32 input -- one step against the update written out with an fp64 draw from the same seed."""
    from azula_amd import sample as S

    monkeypatch.setattr(S, "WIDE_FUSED", fused)
    g = golden("g11_sampler_dtype")
    den = unet_denoiser(g)
    x1 = g["unet_x1"].cuda()
    smp = S.ItoSampler(den, steps=1, start=0.9, stop=0.5, eta=1.0, silent=True, dtype=torch.float64)
    torch.manual_seed(5)
    out = smp(x1)
    torch.manual_seed(5)
    eps = torch.randn(x1.shape, dtype=torch.float64, device="cuda")
    t, s = torch.tensor(0.9, dtype=torch.float64), torch.tensor(0.5, dtype=torch.float64)
    a_t, s_t = den.schedule(t)
    a_s, s_s = den.schedule(s)
    mean = den(x1, t).mean
    r, k, k_eps = smp._ito(a_t, s_t, a_s, s_s)
    ref = r.item() * x1.double() + k.item() * (x1.double() - a_t.item() * mean.double()) + k_eps.item() * eps
    sc = max(1.0, ref.abs().max().item())
    print("Ito, one fp64 step from an fp32 state (fused =", fused, "): max|d|", max_err(out, ref))
    assert out.dtype == torch.float64 and max_err(out, ref) < 1e-5 * sc  # (two evaluations of the fp32 backbone agree to its round-off)
    wrong = r.item() * x1.double() + k.item() * (x1.double() - a_t.item() * mean.double())
    torch.manual_seed(5)
    wrong = wrong + k_eps.item() * torch.randn_like(x1).double()
    assert max_err(out, wrong) > 1e-2  # an fp32 draw from the same seed is a different sample


def test_fp64_elementwise_kernels_bit_exact():
    """az_axpby_f64 / az_scale_f64_to_f32 / az_transition_f64 against torch's fp64 CPU ops, op for op."""
    import ctypes as C

    from azula_amd import _lib
    from azula_amd.denoise import axpby_wide, precondition_wide
    from azula_amd.engine import transition_args

    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 5, 7, 9, generator=g, dtype=torch.float64)
    z32 = torch.randn(3, 5, 7, 9, generator=g)
    a, b = torch.tensor(0.37, dtype=torch.float64), torch.tensor(-1.9, dtype=torch.float64)
    # the denoisers' scalars are (1, ..., 1)-SHAPED fp64 tensors: they promote the fp32 operand before the product
    a4, b4 = a.reshape(1, 1, 1, 1), b.reshape(1, 1, 1, 1)
    assert torch.equal(axpby_wide(a, x.cuda(), b, z32.cuda()).cpu(), a4 * x + b4 * z32)
    ab = torch.randn(3, generator=g, dtype=torch.float64)
    y = axpby_wide(ab, x.cuda(), 2 * ab, x.cuda().flip(0)).cpu()
    assert torch.equal(y, ab[:, None, None, None] * x + (2 * ab)[:, None, None, None] * x.flip(0))
    assert torch.equal(precondition_wide(x.cuda(), a).cpu(), (a4 * x).to(torch.float32))
    n = x.numel()
    m, e = torch.randn(n, generator=g, dtype=torch.float64), torch.randn(n, generator=g, dtype=torch.float64)
    row = torch.zeros(12, dtype=torch.float64)
    row[2], row[4], row[5], row[6], row[7], row[9], row[10] = 1.0, 0.3, 0.8, 0.6, 0.2, -float("inf"), float("inf")
    xs = torch.empty(n, dtype=torch.float64, device="cuda")
    xf, mc, ec, rc = x.flatten().cuda(), m.cuda(), e.cuda(), row.cuda()
    arg = transition_args(x_t=xf.data_ptr(), F=mc.data_ptr(), eps=ec.data_ptr(), x_s=xs.data_ptr(), batch=1, channels=1, inner=n,
                          f_channels=1, coef=rc.data_ptr())
    _lib.call("az_transition_f64", C.byref(arg), _lib.stream_ptr())
    ref = 0.8 * m
    ref = ref + 0.6 * (x.flatten() - 0.3 * m)
    ref = ref + 0.2 * e
    assert torch.equal(xs.cpu(), ref)
