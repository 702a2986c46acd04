r"""Host-side logic of the drop-in API (runs without a GPU).

Covers the reference's CPU-runnable configuration (BASELINE.json configs[0]: KarrasDenoiser +
2-layer MLP + VPSchedule + DDPMSampler(1000) on CPU) against the golden vectors, the
reference's own invariant test (tests/test_denoise.py:135-143) and the host coefficient tables.
"""

import math
import os

import pytest
import torch

from conftest import max_err
from azula_amd.denoise import DiracPosterior, GaussianPosterior, KarrasDenoiser
from azula_amd.noise import Schedule, VESchedule, VPSchedule
from azula_amd.sample import DDIMSampler, DDPMSampler
from oracle import sampling, synth


class ToyMLP(torch.nn.Module):
    """Same architecture as the reference tests' Dummy (tests/test_sample.py:28-53)."""

    def __init__(self, features=5):
        super().__init__()
        self.l1 = torch.nn.Linear(features, 64)
        self.l2 = torch.nn.Linear(64, features)

    def forward(self, x_t, t, label=None):
        f = torch.exp(torch.log(torch.tensor(1e-4)) * torch.linspace(0, 1, 32, dtype=x_t.dtype))
        e = torch.cat((torch.sin(t.unsqueeze(-1) * f), torch.cos(t.unsqueeze(-1) * f)), dim=-1)
        return self.l2(torch.relu(self.l1(x_t) + e))


def toy_denoiser(g):
    net = ToyMLP()
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    return KarrasDenoiser(net, VPSchedule()).eval()


def test_schedule_tables_match_reference(golden):
    g = golden("g1_schedule")
    for case in g.meta["cases"]:
        sched = VPSchedule(case["alpha_min"], case["sigma_min"])
        ts = DDIMSampler(None, steps=case["steps"]).timesteps
        assert torch.equal(ts, g[case["tag"] + "_t"])
        al, si = zip(*(sched(t) for t in ts.unbind()))
        torch.testing.assert_close(torch.stack(al), g[case["tag"] + "_alpha"], rtol=2e-7, atol=1e-9)
        torch.testing.assert_close(torch.stack(si), g[case["tag"] + "_sigma"], rtol=2e-7, atol=1e-9)
    assert isinstance(VPSchedule(), Schedule)
    a, s = VESchedule()(torch.tensor(0.0))
    assert a == 1 and abs(s.item() - 1e-3) < 1e-9


def test_config1_readme_cpu(golden):
    g = golden("g4_toy_loop")
    den = toy_denoiser(g)
    torch.manual_seed(g.meta["init_seed"])
    smp = DDPMSampler(den, steps=1000, silent=True)
    x1 = smp.init((64, 5))
    assert torch.equal(x1, g["x1"])
    torch.manual_seed(g.meta["loop_seed"])
    x0 = smp(x1)
    assert x0.shape == (64, 5) and torch.isfinite(x0).all()
    assert max_err(x0, g["ddpm1000"]) < 1e-4
    x0 = DDIMSampler(den, steps=64, silent=True)(x1)
    assert max_err(x0, g["ddim64"]) < 1e-5


@pytest.mark.parametrize("batch", [(), (64,)])
def test_ve_reschedule_invariance(batch):
    """Reference tests/test_denoise.py:135-143: the mean is unchanged by re-scheduling to VE."""

    class ReSchedule(Schedule):
        def __init__(self, s):
            self.s = s

        def __call__(self, t):
            a, s = self.s(t)
            return torch.ones_like(a), s / a

    den = KarrasDenoiser(ToyMLP(), VPSchedule())
    x = torch.randn(*batch, 5)
    t = torch.rand(batch)
    a, s = den.schedule(t)
    x_t = torch.normal(a[..., None] * x, s[..., None])
    q = den(x_t, t)
    assert isinstance(q, DiracPosterior) and q.mean.shape == x.shape
    den.schedule = ReSchedule(den.schedule)
    q_ve = den(x_t / a[..., None], t)
    assert torch.allclose(q.mean, q_ve.mean, atol=1e-6)


def test_ddim_eta1_equals_ddpm(golden):
    g = golden("g4_toy_loop")
    den = toy_denoiser(g)
    torch.manual_seed(7)
    a = DDPMSampler(den, steps=32, silent=True)(g["x1"])
    torch.manual_seed(7)
    b = DDIMSampler(den, eta=1.0, steps=32, silent=True)(g["x1"])
    assert max_err(a, b) < 1e-6


def test_host_coefficient_table_matches_golden(golden):
    """The (steps, 16) AzStepCoef table the fused path uploads, against G2/G1."""
    from azula_amd._lib import COEF_FIELDS
    from azula_amd.sample import FusedDenoiser

    g = golden("g2_precond")
    den = KarrasDenoiser(torch.nn.Identity(), VPSchedule())
    smp = DDIMSampler(den, steps=64)
    tab = smp._host_table(FusedDenoiser(coefficients=den.host_coefficients, programs=[]))
    col = {n: i for i, n in enumerate(COEF_FIELDS)}
    assert tab.shape == (64, 16)
    for j, name in enumerate(["c_in", "c_out", "c_skip", "c_time"]):
        torch.testing.assert_close(tab[:, col[name]], g["karras"][:64, j], rtol=2e-7, atol=1e-9)
    torch.testing.assert_close(tab[:-1, col["c_in_next"]], tab[1:, col["c_in"]], rtol=0, atol=0)
    assert tab[-1, col["c_in_next"]] == 0
    assert (tab[:, col["k_eps"]] == 0).all()  # eta = 0
    assert tab.view(torch.int32)[:, col["step"]].tolist() == list(range(64))
    tab = DDPMSampler(den, steps=64)._host_table(FusedDenoiser(coefficients=den.host_coefficients, programs=[]))
    assert (tab[:, col["k_eps"]] > 0).all()


def test_gaussian_posterior_log_prob():
    """Reference tests/test_denoise.py:48-65."""
    mean, var, x = torch.randn(7), torch.rand(7) + 0.1, torch.randn(7)
    ref = torch.distributions.Normal(mean, var.sqrt()).log_prob(x)
    assert torch.allclose(GaussianPosterior(mean, var).log_prob(x), ref, atol=1e-6)


def test_unet_state_dict_matches_reference_shapes(golden):
    from azula_amd.nn import UNet

    for name in ("unet_group", "unet_layer_odd", "unet_rms_nomod"):
        g = golden("g5_" + name)
        cfg = g.meta["cfg"]
        net = UNet(
            cfg["in_channels"], cfg["out_channels"], hid_channels=cfg["hid_channels"], hid_blocks=cfg["hid_blocks"],
            norm=cfg["norm"], groups=cfg["groups"], mod_features=cfg["mod_features"],
        )
        mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        ref = {k: tuple(v) for k, v in g.meta["shapes"].items()}
        assert mine == ref, name


def test_backbones_fail_loudly_on_cpu():
    from azula_amd.nn import TimeModulated, UNet

    net = UNet(3, 3, hid_channels=(8, 16), hid_blocks=(1, 1), mod_features=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.randn(1, 3, 8, 8), torch.randn(8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TimeModulated(net, 8)(torch.randn(1, 3, 8, 8), torch.tensor(0.3))


def test_vit_and_adm_state_dicts_match_reference_shapes(golden):
    from azula_amd.nn import ViT
    from azula_amd.plugins import adm

    g = golden("g5_vit")
    cfg = g.meta["cfg"]
    net = ViT(cfg["in_channels"], cfg["out_channels"], hid_channels=cfg["hid_channels"], hid_blocks=cfg["hid_blocks"],
              attention_heads=cfg["attention_heads"], patch_size=cfg["patch_size"], mod_features=cfg["mod_features"])
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in g.meta["shapes"].items()}
    for name in ("adm_uncond", "adm_cond_neworder"):
        g = golden("g5_" + name)
        den = adm.make_model(**g.meta["cfg"])
        assert {k: tuple(v.shape) for k, v in den.backbone.state_dict().items()} == {
            k: tuple(v) for k, v in g.meta["shapes"].items()
        }
        torch.testing.assert_close(den.sigmas, golden("g2_precond")[
            "adm_sigmas" if g.meta["cfg"]["discrete_schedule"] == "linear" else "adm_sigmas_cosine"], rtol=2e-7, atol=1e-9)
    # zero-init quirk of the reference is reproduced: random-init eps-hat == 0
    den = adm.make_model(**golden("g5_adm_uncond").meta["cfg"])
    assert not den.backbone.out[2].weight.any() and not den.backbone.middle_block[1].proj_out.weight.any()


def test_adm_cards_and_hub_layout(tmp_path):
    from azula_amd import hub
    from azula_amd.plugins import adm, load_cards

    cards = load_cards(adm)
    assert list(cards) == ["imagenet_64x64_cond", "imagenet_128x128_cond", "imagenet_256x256", "imagenet_256x256_cond",
                           "imagenet_512x512_cond", "ffhq_256x256"]
    c = cards["imagenet_256x256"]
    assert c.config["num_channels"] == 256 and c.config["channel_mult"] == [1, 1, 2, 2, 4, 4] and c.config["num_classes"] is None
    hub.set_hub_dir(str(tmp_path))
    assert hub.cached_path(c.url).endswith("https.openaipublic.blob.core.windows.net.diffusion.jul.2021.256x256_diffusion_uncond.pt")
    with pytest.raises(FileNotFoundError, match="does not download"):
        adm.load_model("imagenet_256x256")
    # adm_coefficients on the host: DDIM-64 indices of the discrete table
    den = adm.AblatedDenoiser(torch.nn.Identity())
    a, s = den.schedule(torch.tensor(1.0))
    co = den.host_coefficients(a, s)
    assert int(co["time_index"]) == 954 and abs(float(co["c_out"]) + 100) < 1.0


def test_next_samplers_cpu_match_reference(golden):
    """Euler / Heun / Ito on the host follow the reference bit for bit (G8)."""
    from azula_amd.sample import EulerSampler, HeunSampler, ItoSampler

    g = golden("g8_toy_next_samplers")
    net = ToyMLP()
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    den = KarrasDenoiser(net, VPSchedule()).eval()
    assert max_err(EulerSampler(den, steps=64, silent=True)(g["x1"]), g["euler64"]) < 1e-5
    assert max_err(HeunSampler(den, steps=64, silent=True)(g["x1"]), g["heun64"]) < 1e-5
    torch.manual_seed(2)
    x0 = ItoSampler(den, steps=64, silent=True, **g.meta["ito"])(g["x1"])
    assert max_err(x0, g["ito64"]) < 1e-5
    # DDIM(eta=0) and Euler are the same ODE step (reference docstring sample.py:236-237)
    assert max_err(DDIMSampler(den, steps=64, silent=True)(g["x1"]), g["euler64"]) < 1e-4


MULTISTEP = ("zAB", "vAB", "zEAB", "xEAB", "REAB")


def _emulate_multistep_kernel(smp, den, x):
    """az_multistep_f32's arithmetic (separately rounded fp32 mul/add, history oldest first) in torch,
    driven by the sampler's own device coefficient table."""
    alpha, sigma = den.schedule(smp.timesteps)
    table = smp._device_table(alpha, sigma)
    x_t, hist = x, []
    for i, t in enumerate(smp.timesteps[:-1].unbind()):
        a, b, p, w_new = table[i, :4]
        mean = den(x_t, t).mean
        pred = a * x_t + b * mean
        acc = p * x_t
        for j, h in enumerate(hist):
            acc = acc + table[i, 4 + j] * h
        x_t = acc + w_new * pred
        hist = (hist + [pred])[-(smp.order - 1) :] if smp.order > 1 else []
    return x_t


def test_multistep_samplers_cpu_match_reference(golden):
    """the Adams-Bashforth family and the PC sampler.  (1) The host path reproduces the
    reference's vectors (G9); (2) the folded (a, b, p, w) table that drives az_multistep_f32 gives the
    same samples when the kernel's arithmetic is emulated in fp32 -- the part of the device path that
    can be checked without a GPU."""
    import azula_amd.sample as S

    g = golden("g9_toy_multistep")
    net = ToyMLP()
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"]))
    den = KarrasDenoiser(net, VPSchedule()).eval()
    steps = g.meta["steps"]
    for kind in MULTISTEP:
        for order in (1, 2, 3):
            smp = getattr(S, kind + "Sampler")(den, order=order, steps=steps, silent=True)
            want = g[f"{kind}_o{order}"]
            assert max_err(smp(g["x1"]), want) < 2e-5, (kind, order)
            assert max_err(_emulate_multistep_kernel(smp, den, g["x1"]), want) < 2e-4, (kind, order)
    torch.manual_seed(3)
    x0 = S.PCSampler(den, steps=steps, silent=True, **g.meta["pc"])(g["x1"])
    assert max_err(x0, g["pc"]) < 1e-5
    with pytest.raises(ValueError, match="order"):
        S.zABSampler(den, order=9, steps=4, silent=True)(g["x1"])


def test_multistep_weight_functions_keep_reference_names(golden):
    """The reference exposes the solves as static methods; callers (and its tests) use them."""
    import azula_amd.sample as S

    g = golden("g9_multistep_weights")
    alpha, sigma = g["alpha"], g["sigma"]
    u = sigma.log() - alpha.log()
    for kind, fn in (("zEAB", S.zEABSampler._exponential_adams_bashforth), ("xEAB", S.xEABSampler._exponential_adams_bashforth),
                     ("REAB", S.REABSampler._exponential_adams_bashforth)):
        c = fn(u, i=7, n=3)
        assert c.dtype == torch.float32
        torch.testing.assert_close(c, g[f"{kind}_w3"][7, :3], rtol=2e-5, atol=1e-7)
    c = S.vABSampler._adams_bashforth(sigma / (alpha + sigma), i=1, n=4)  # n is clipped to i + 1
    torch.testing.assert_close(c, g["vAB_w4"][1, :2], rtol=2e-5, atol=1e-7)


def test_schedules_and_simple_denoiser_host():
    """the remaining closed-form schedules (azula/noise.py:132-231) and SimpleDenoiser
    (azula/denoise.py:177-230) against the oracle formulas; fused coefficients are Karras with c_skip 0."""
    from azula_amd.denoise import SimpleDenoiser
    from azula_amd.noise import CosineSchedule, DecaySchedule, RectifiedSchedule

    t = torch.linspace(0, 1, 17)
    a, s = RectifiedSchedule(1e-2, 2e-3)(t)
    oa, os_ = sampling.rectified_schedule(t, 1e-2, 2e-3)
    assert torch.equal(a, oa) and torch.equal(s, os_)
    a, s = CosineSchedule()(t)
    assert torch.equal(a, torch.cos(math.acos(1e-3) * t)) and torch.equal(s, torch.sqrt(1 - a**2 + 1e-3**2))
    a, s = DecaySchedule(gamma=0.3)(t)
    tau = (1 - 0.3**t) / (1 - 0.3)
    assert torch.equal(a, tau * 1e-3 + (1 - tau)) and torch.equal(s, tau + (1 - tau) * 1e-3)
    assert a[0] == 1 and abs(float(a[-1]) - 1e-3) < 1e-6  # endpoints: clean at t = 0, alpha_min at t = 1

    net = ToyMLP()
    den = SimpleDenoiser(net, CosineSchedule()).eval()
    x, tt = torch.randn(4, 5), torch.tensor(0.3)
    al, sg = den.schedule(tt)
    want = net(torch.rsqrt(al**2 + sg**2) * x, torch.log(sg / al))
    assert torch.equal(den(x, tt).mean, want)
    co = den.host_coefficients(al, sg)
    assert float(co["c_skip"]) == 0 and float(co["c_out"]) == 1 and torch.equal(co["c_time"], torch.log(sg / al))


def test_jit_plugin_host_side(golden):
    """JiT plugin without a GPU: cards, state_dict compatibility with the reference's parameter shapes (G10),
    JITDenoiser coefficients, and the loud failure of the backbone on CPU tensors."""
    from azula_amd.plugins import jit
    from azula_amd.plugins.jit.model import rotary_tables, sincos_table
    from azula_amd.plugins.utils import load_cards
    from oracle import nets

    cards = load_cards(jit)
    assert set(cards) == {f"jit_{s}_{p}" for s in ("0.1b", "0.5b", "1.0b") for p in (16, 32)}
    assert cards["jit_0.5b_32"].config == {"model": "JiT-L/32"}
    with pytest.raises(FileNotFoundError, match="does not download"):
        jit.load_model("jit_0.1b_16")
    jit.JiT(input_size=32, patch_size=4, hidden_size=96, depth=1, num_heads=2)  # head_dim 48: zero-padded to 64 (round 6)
    with pytest.raises(NotImplementedError, match="head_dim"):
        jit.JiT(input_size=32, patch_size=4, hidden_size=288, depth=1, num_heads=2)  # head_dim 144 > 128: no kernel

    g = golden("g10_jit_ctx")
    net = jit.JiT(**g.meta["cfg"])
    shapes = {k: tuple(v) for k, v in g.meta["shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == shapes
    net.load_state_dict(synth.synth_state_dict(shapes, g.meta["weight_seed"]))
    den = jit.JITDenoiser(net, num_classes=g.meta["cfg"]["num_classes"])
    al, sg = den.schedule(torch.tensor(0.25))
    co = den.host_coefficients(al, sg)
    assert torch.equal(co["c_in"], 1 / (al + sg)) and torch.equal(co["c_time"], al / (al + sg))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        den(g["x"], torch.tensor(0.4))
    # build-time tables == the reference's buffers (restated in the oracle, pinned by make_golden)
    assert torch.equal(sincos_table(64, 8), nets.jit_pos_embed(64, 8))
    cos, sin = rotary_tables(16, 4, 8, 8)
    ocos, osin = nets.jit_rope_tables(16, 8, 8)
    assert torch.equal(cos[:, 2], ocos[:, 0::2]) and torch.equal(sin[:, 0], osin[:, 1::2])


def test_layers_for_custom_backbones():
    """azula_amd.nn.layers (counterpart of azula.nn.layers) against the oracle formulas."""
    from azula_amd.nn import layers
    from oracle import nets

    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 8, 12, generator=g)
    assert torch.equal(layers.LayerNorm(dim=-3)(x), nets.layer_norm_unbiased(x, dim=-3))
    assert torch.equal(layers.RMSNorm(dim=1)(x), nets.rms_norm(x, dim=1))
    t = torch.rand(7, generator=g)
    assert torch.equal(layers.SineEncoding(64)(t), nets.sine_encoding(t, 64))
    assert layers.SineEncoding(32)(x.half()).dtype == torch.float16  # promote_dtype casts back
    for cl in (False, True):
        p = layers.Patchify((2, 3), channel_last=cl)(x)
        assert p.shape == ((2, 4, 4, 36) if cl else (2, 36, 4, 4))
        assert torch.equal(layers.Unpatchify((2, 3), channel_last=cl)(p), x)
    # '... Z (A a) (B b) -> ... A B (Z a b)': feature index z*p*p + a*p + b
    p = layers.Patchify((2, 2), channel_last=True)(x)
    assert p[1, 2, 3, 1 * 4 + 1 * 2 + 0] == x[1, 1, 2 * 2 + 1, 3 * 2 + 0]
    y = torch.randn(4, 10, generator=g)
    assert torch.equal(layers.SwiGLU()(y), y[:, 0::2] * torch.nn.functional.silu(y[:, 1::2]))
    assert torch.equal(layers.ReLU2()(y), torch.relu(y) ** 2)
    conv = layers.ConvNd(3, 5, spatial=2, identity_init=True, kernel_size=3, padding=1)
    assert abs(conv.weight[1, 1, 1, 1].item() - 1) < 0.1 and conv.weight[1, 0].abs().max() < 0.1


@pytest.mark.parametrize("with_label", [False, True])
@pytest.mark.parametrize("batch", [(), (64,)])
def test_samplers_shapes_and_kwargs(with_label, batch):
    """Mirror of the reference's tests/test_sample.py:57-92, all 12 sampler configurations:
    init / output shapes, finiteness, and keyword arguments flowing untouched to the backbone."""
    from functools import partial

    from azula_amd.nn.layers import SineEncoding
    from azula_amd.sample import (EulerSampler, HeunSampler, ItoSampler, PCSampler, REABSampler, vABSampler, xEABSampler,
                                  zABSampler, zEABSampler)

    class Dummy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.l2 = torch.nn.Linear(5, 64), torch.nn.Linear(64, 5)
            self.time_encoding = SineEncoding(64)

        def forward(self, x_t, t, label=None):
            assert isinstance(label, str) if with_label else label is None
            return self.l2(torch.relu(self.l1(x_t) + self.time_encoding(t)))

    den = KarrasDenoiser(Dummy(), VPSchedule())
    for S in (partial(DDPMSampler), partial(DDIMSampler, eta=0.0), partial(DDIMSampler, eta=1.0), partial(EulerSampler),
              partial(HeunSampler), partial(ItoSampler, eta=1.0), partial(zABSampler), partial(vABSampler),
              partial(zEABSampler), partial(xEABSampler), partial(REABSampler), partial(PCSampler, corrections=1)):
        sampler = S(den, steps=64, silent=True)
        x1 = sampler.init((*batch, 5))
        assert x1.shape == (*batch, 5) and torch.isfinite(x1).all()
        x0 = sampler(x1, label="cat") if with_label else sampler(x1)
        assert x0.shape == (*batch, 5) and torch.isfinite(x0).all()


@pytest.mark.parametrize("batch", [(), (64,)])
def test_schedule_properties(batch):
    """Mirror of the reference's tests/test_noise.py:11-43 for the five closed-form schedules: positive scales,
    signal-to-noise ratio non-increasing in t, alpha_0 == 1."""
    from azula_amd.noise import CosineSchedule, DecaySchedule, RectifiedSchedule

    torch.manual_seed(0)
    for S in (VPSchedule, VESchedule, CosineSchedule, RectifiedSchedule, DecaySchedule):
        schedule = S()
        assert isinstance(schedule, Schedule)
        t = torch.rand(batch)
        alpha_t, sigma_t = schedule(t)
        assert alpha_t.shape == batch and sigma_t.shape == batch, S
        assert (alpha_t > 0).all() and (sigma_t > 0).all(), S
        s = torch.rand_like(t) * t
        alpha_s, sigma_s = schedule(s)
        assert (alpha_s / sigma_s >= alpha_t / sigma_t).all(), S
        assert (schedule(torch.zeros(()))[0] == 1).all(), S


def test_hub_cache_semantics(tmp_path):
    """azula/hub.py:34-124 without the network: sanitised cache names, optional "alg:prefix" hash check, archive
    extraction into "<file>+x" (what jit.load_model joins "checkpoint-last.pth" onto)."""
    import hashlib
    import zipfile

    from azula_amd import hub

    old = hub.get_hub_dir()
    try:
        hub.set_hub_dir(str(tmp_path))
        url = "https://example.org/some dir/jit-b-16?rlkey=abc&dl=1"
        path = hub.cached_path(url)
        assert os.path.basename(path) == "https.example.org.some.dir.jit.b.16.rlkey.abc.dl.1"
        with pytest.raises(FileNotFoundError, match="does not download"):
            hub.download(url, quiet=True)
        with zipfile.ZipFile(path, "w") as z:
            z.writestr("checkpoint-last.pth", b"weights")
        digest = hashlib.sha256(open(path, "rb").read()).hexdigest()
        assert hub.download(url, hash_prefix="sha256:" + digest[:10], quiet=True) == path
        with pytest.raises(AssertionError, match="does not match"):
            hub.download(url, hash_prefix="sha256:0000", quiet=True)
        xd = hub.download(url, extract=True, quiet=True)
        assert xd == path + "+x" and open(os.path.join(xd, "checkpoint-last.pth"), "rb").read() == b"weights"
        assert hub.download(url, extract=True, quiet=True) == xd  # second call: already unpacked
        explicit = tmp_path / "elsewhere.bin"
        explicit.write_bytes(b"x")
        assert hub.download("ignored://", filename=str(explicit), quiet=True) == str(explicit)
    finally:
        hub.set_hub_dir(old)


def test_load_model_from_a_populated_hub_cache(tmp_path, monkeypatch):
    """SURVEY 8f.4 weight I/O: adm.load_model / jit.load_model read a checkpoint that sits in the reference's hub cache
    layout (no download).  Tiny stand-in cards keep the files small; the checkpoints use the reference's key patterns
    (guided-diffusion state_dict; JiT's {"model_ema1": {"net.<key>": ...}} inside an extracted archive)."""
    import zipfile
    from types import SimpleNamespace

    from azula_amd import hub
    from azula_amd.plugins import adm, jit

    old = hub.get_hub_dir()
    hub.set_hub_dir(str(tmp_path))
    try:
        # ---- ADM
        cfg = dict(image_size=32, num_channels=32, num_res_blocks=1, channel_mult=(1, 2), attention_resolutions=(16,),
                   num_heads=2, num_head_channels=-1, resblock_updown=True, use_scale_shift_norm=True)
        url = "https://example.org/diffusion/tiny_adm.pt"
        src = adm.make_model(**cfg)
        for p in src.backbone.parameters():
            p.data.normal_()
        torch.save(src.backbone.state_dict(), hub.cached_path(url))
        monkeypatch.setattr(adm, "load_cards", lambda _: {"tiny": SimpleNamespace(url=url, hash=None, config=cfg)})
        den = adm.load_model("tiny")
        assert not den.training
        for (k, a), (_, b) in zip(src.backbone.state_dict().items(), den.backbone.state_dict().items()):
            assert torch.equal(a, b), k
        # ---- JiT: archive with checkpoint-last.pth holding EMA weights under a "net." prefix
        jcfg = dict(input_size=32, patch_size=4, hidden_size=64, depth=2, num_heads=4, bottleneck_dim=16, in_context_len=4,
                    in_context_start=1, num_classes=10)
        jurl = "https://example.org/jit/tiny?dl=1"
        jsrc = jit.JiT(**jcfg)
        for p in jsrc.parameters():
            p.data.normal_()
        ckpt = tmp_path / "checkpoint-last.pth"
        torch.save({"model_ema1": {"net." + k: v for k, v in jsrc.state_dict().items()}, "model": {}}, ckpt)
        with zipfile.ZipFile(hub.cached_path(jurl), "w") as z:
            z.write(ckpt, "checkpoint-last.pth")
        monkeypatch.setattr(jit, "load_cards", lambda _: {"tiny": SimpleNamespace(url=jurl, hash=None, config={"model": "tiny"})})
        monkeypatch.setitem(jit.JiT_models, "tiny", lambda **kw: jit.JiT(**jcfg, **kw))
        jden = jit.load_model("tiny")
        assert isinstance(jden, jit.JITDenoiser) and jden.num_classes == 10
        for (k, a), (_, b) in zip(jsrc.state_dict().items(), jden.backbone.state_dict().items()):
            assert torch.equal(a, b), k
    finally:
        hub.set_hub_dir(old)


def test_sampler_dtype_float64_promotes_the_latents_like_the_reference(golden):
    """G11: ``Sampler(dtype=float64)`` -- an fp64 time grid; the (1, ..., 1)-shaped schedule scalars promote the fp32
    latents, x0 is fp64 (azula/sample.py:69-94, azula/denoise.py:306-322).  Host path: the reference's op sequence."""
    g = golden("g11_sampler_dtype")
    net = ToyMLP()
    net.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in g.meta["toy_shapes"].items()}, g.meta["toy_weight_seed"]))
    den = KarrasDenoiser(net, VPSchedule()).eval()
    torch.manual_seed(g.meta["loop_seed"])
    x0 = DDIMSampler(den, steps=16, silent=True, dtype=torch.float64)(g["toy_x1"])
    assert x0.dtype == torch.float64
    torch.testing.assert_close(x0, g["toy_ddim16"], rtol=1e-12, atol=1e-12)
    torch.manual_seed(g.meta["loop_seed"])
    x0 = DDPMSampler(den, steps=16, silent=True, dtype=torch.float64)(g["toy_x1"])
    torch.testing.assert_close(x0, g["toy_ddpm16"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("order", ["nHC", "H3C", "3HC"])
@pytest.mark.parametrize("d,dp", [(24, 32), (48, 64), (96, 128), (20, 32)])
def test_zero_padded_heads_leave_attention_unchanged(order, d, dp):
    """engine.pad_qkv_heads / pad_proj_heads (host arithmetic, round 6): a head of d channels zero-padded to the next instantiated
    size d' -- projections packed with zero rows / columns per head, scale 1 / sqrt(d) -- computes the same attention (here in torch
    on the CPU, in the three q | k | v channel orders of azula/nn/attention.py:90 and plugins/adm/_src/unet.py:338,371)."""
    import torch.nn.functional as F

    from azula_amd import engine

    assert engine.attn_padded_dim(d) == dp and engine.attn_padded_dim(64) == 64 and engine.attn_padded_dim(8) == 8
    with pytest.raises(NotImplementedError):
        engine.attn_padded_dim(129)
    g = torch.Generator().manual_seed(d)
    B, L, H, C = 2, 10, 3, 3 * d
    x = torch.randn(B, L, C, generator=g)
    wq, bq = torch.randn(3 * C, C, generator=g) / C**0.5, torch.randn(3 * C, generator=g)
    wo = torch.randn(C, C, generator=g) / C**0.5

    def attend(w, b, wout, dd, scale):
        qkv = F.linear(x, w, b)
        if order in ("nHC", "3HC"):
            q, k, v = qkv.reshape(B, L, 3, H, dd).permute(2, 0, 3, 1, 4)
        else:
            q, k, v = qkv.reshape(B, L, H, 3, dd).permute(3, 0, 2, 1, 4)
        y = F.scaled_dot_product_attention(q, k, v, scale=scale)
        return F.linear(y.transpose(1, 2).reshape(B, L, H * dd), wout)

    ref = attend(wq, bq, wo, d, d**-0.5)
    wp, bp = engine.pad_qkv_heads(wq, bq, H, d, dp, order)
    got = attend(wp, bp, engine.pad_proj_heads(wo, H, d, dp), dp, d**-0.5)
    assert wp.shape == (3 * H * dp, C) and torch.allclose(got, ref, atol=1e-5, rtol=1e-5)
    t = engine.pad_head_table(torch.arange(2 * H * (d // 2), dtype=torch.float32).reshape(2, -1), H, d // 2, dp // 2)
    assert t.shape == (2, H * (dp // 2)) and t.reshape(2, H, dp // 2)[..., d // 2 :].abs().sum() == 0
