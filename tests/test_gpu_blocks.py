r"""The corners of the hot-path modules' API on the GPU, against reference-generated vectors (G12): standalone
``MultiheadSelfAttention`` / ``DiTBlock`` / ``UNetBlock`` forwards (one-block plans), the boolean attention ``mask``
(azula/nn/attention.py:72-104), ViT ``cond`` / ``unpatch_size`` (azula/nn/vit.py:40-106) and UNet ``periodic=True``
(circular padding in the conv gathers, azula/nn/unet.py:175-180)."""

import pytest
import torch

from conftest import max_err
from oracle import nets, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
NAME = "g12_blocks_mask_cond_periodic"


def shapes(g, key):
    return {n: tuple(v) for n, v in g.meta[key].items()}


def test_standalone_attention_with_masks(golden):
    from azula_amd.nn import MultiheadSelfAttention

    g = golden(NAME)
    msa = MultiheadSelfAttention(64, pos_channels=2, attention_heads=4, rope=True)
    msa.load_state_dict(synth.synth_state_dict(shapes(g, "msa_shapes"), 31))
    msa = msa.cuda().eval()
    x, pos = g["msa_x"].cuda(), g["msa_pos"].cuda()
    for tag, mask in (("nomask", None), ("causal", g["msa_causal"].bool().cuda()), ("bmask", g["msa_bmask"].bool().cuda())):
        y = msa(x, pos, mask)
        err = max_err(y, g["msa_y_" + tag])
        print("standalone MSA", tag, "max|d|", err, "scale", g["msa_y_" + tag].abs().max().item())
        assert y.shape == x.shape and err < 2e-5 * max(1.0, g["msa_y_" + tag].abs().max().item())
    # leading dims other than one batch axis: (2, 1, L, C)
    y4 = msa(x[:, None], pos, g["msa_causal"].bool().cuda())
    assert y4.shape == (2, 1, 9, 64) and max_err(y4[:, 0], g["msa_y_causal"]) < 2e-5
    # a fully masked query row is NaN, as in the reference's softmax over -inf
    dead = g["msa_causal"].bool().clone()
    dead[3] = False
    y = msa(x, pos, dead.cuda())
    assert torch.isnan(y[:, 3]).all() and torch.isfinite(y[:, [0, 1, 2, 4]]).all()


def test_attention_mask_through_the_half_precision_kernel(golden):
    from azula_amd.nn import MultiheadSelfAttention

    g = golden(NAME)
    msa = MultiheadSelfAttention(64, pos_channels=2, attention_heads=4, rope=True)
    msa.load_state_dict(synth.synth_state_dict(shapes(g, "msa_shapes"), 31))
    msa = msa.cuda().eval().bfloat16()
    y = msa(g["msa_x"].cuda(), g["msa_pos"].cuda(), g["msa_causal"].bool().cuda())
    ref = g["msa_y_causal"]
    assert y.dtype == torch.float32 and max_err(y, ref) < 3e-2 * max(1.0, ref.abs().max().item())


def test_standalone_dit_block(golden):
    from azula_amd.nn import DiTBlock

    g = golden(NAME)
    blk = DiTBlock(64, mod_features=16, pos_channels=2, attention_heads=4, rope=True, ffn_activation="swiglu")
    blk.load_state_dict(synth.synth_state_dict(shapes(g, "dit_shapes"), 32))
    blk = blk.cuda().eval()
    x, pos, mask, mod = g["msa_x"].cuda(), g["msa_pos"].cuda(), g["msa_causal"].bool().cuda(), g["dit_mod"].cuda()
    sc = max(1.0, g["dit_y"].abs().max().item())
    y = blk(x, mod, pos, mask)
    print("standalone DiTBlock max|d|", max_err(y, g["dit_y"]), "scale", sc)
    assert max_err(y, g["dit_y"]) < 2e-5 * sc
    assert max_err(blk(x, mod[0], pos, mask), g["dit_y_mod1"]) < 2e-5 * sc  # mod of shape (D)
    assert torch.equal(blk(x, mod, pos, mask), y)  # the cached one-block plan replays deterministically
    blk.ffn[3].weight.mul_(2.0)  # parameter update (version counter bumps): the plan is rebuilt
    assert not torch.equal(blk(x, mod, pos, mask), y)


def test_standalone_unet_block(golden):
    from azula_amd.nn import UNetBlock

    g = golden(NAME)
    ub = UNetBlock(12, mod_features=16, norm="group", groups=4, spatial=2, kernel_size=3, padding=1)
    ub.load_state_dict(synth.synth_state_dict(shapes(g, "ublock_shapes"), 33))
    ub = ub.cuda().eval()
    y = ub(g["ublock_x"].cuda(), g["mod"].cuda())
    sc = max(1.0, g["ublock_y"].abs().max().item())
    print("standalone UNetBlock max|d|", max_err(y, g["ublock_y"]), "scale", sc)
    assert max_err(y, g["ublock_y"]) < 2e-5 * sc
    ub2 = UNetBlock(6, mod_features=0, norm="layer", spatial=2, kernel_size=3, padding=1)
    ub2.load_state_dict(synth.synth_state_dict(shapes(g, "ublock2_shapes"), 34))
    y2 = ub2.cuda().eval()(g["ublock2_x"].cuda())
    assert max_err(y2, g["ublock2_y"]) < 2e-5 * max(1.0, g["ublock2_y"].abs().max().item())


def test_vit_cond_and_unpatch_size(golden):
    from azula_amd.nn import ViT

    g = golden(NAME)
    vit = ViT(**g.meta["vit_cfg"])
    vit.load_state_dict(synth.synth_state_dict(shapes(g, "vit_shapes"), 35))
    vit = vit.cuda().eval()
    y = vit(g["vit_x"].cuda(), g["mod"].cuda(), g["vit_cond"].cuda())
    sc = max(1.0, g["vit_y"].abs().max().item())
    print("ViT cond + unpatch 1 max|d|", max_err(y, g["vit_y"]), "scale", sc)
    assert y.shape == (2, 2, 4, 6) and max_err(y, g["vit_y"]) < 2e-5 * sc
    with pytest.raises(AssertionError):
        vit(g["vit_x"].cuda(), g["mod"].cuda())  # built with cond_channels: cond is required
    vit3 = ViT(**g.meta["vit3_cfg"])
    vit3.load_state_dict(synth.synth_state_dict(shapes(g, "vit3_shapes"), 36))
    y3 = vit3.cuda().eval()(g["vit3_x"].cuda(), g["mod"][0].cuda())
    assert y3.shape == (1, 3, 10, 8) and max_err(y3, g["vit3_y"]) < 2e-5 * max(1.0, g["vit3_y"].abs().max().item())


@pytest.mark.parametrize("policy", ["1", "2", "0"])
def test_periodic_unet(golden, policy, monkeypatch):
    """Circular padding through the Winograd, the direct and the narrow-output (image head) kernels."""
    from azula_amd import engine
    from azula_amd.nn import UNet

    g = golden(NAME)
    monkeypatch.setattr(engine, "WINOGRAD", policy)
    net = UNet(**g.meta["punet_cfg"], periodic=True)
    net.load_state_dict(synth.synth_state_dict(shapes(g, "punet_shapes"), 37))
    net = net.cuda().eval()
    for name in ("punet_a", "punet_b", "punet_c"):
        x = g[name + "_x"].cuda()
        y = net(x, g["mod"][: x.shape[0]].cuda())
        sc = max(1.0, g[name + "_y"].abs().max().item())
        print("periodic UNet", name, "policy", policy, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
        assert max_err(y, g[name + "_y"]) < 2e-5 * sc
    # a 64-channel periodic layer on a map large enough for full Winograd tile blocks, against the oracle
    torch.manual_seed(0)
    big = UNet(3, 3, hid_channels=(64, 64), hid_blocks=(1, 1), norm="group", groups=8, mod_features=16, periodic=True)
    sd = synth.synth_state_dict(synth.shapes_of(big.state_dict()), seed=9)
    big.load_state_dict(sd)
    x, mod = torch.randn(2, 3, 40, 36), torch.randn(2, 16)
    ref = nets.unet_forward(sd, dict(hid_channels=(64, 64), hid_blocks=(1, 1), norm="group", groups=8, periodic=True), x, mod)
    out = big.cuda().eval()(x.cuda(), mod.cuda())
    assert max_err(out, ref) < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("name", ["unet1d", "unet1d_odd", "unet1d_periodic"])
def test_unet_on_1d_signals(golden, name):
    """``spatial = 1`` (azula/nn/unet.py:119-203): (B, C, L) signals run as one-row images through the same conv /
    GroupNorm kernels; state_dict shapes are the reference's Conv1d ones."""
    from azula_amd.nn import UNet

    g = golden("g13_spatial")
    cfg = dict(g.meta[name + "_cfg"])
    periodic = cfg.pop("periodic")
    net = UNet(**cfg, spatial=1, periodic=periodic)
    sh = {n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == sh
    net.load_state_dict(synth.synth_state_dict(sh, 41))
    y = net.cuda().eval()(g[name + "_x"].cuda(), g["mod"].cuda())
    sc = max(1.0, g[name + "_y"].abs().max().item())
    print(name, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
    assert y.shape == g[name + "_y"].shape and max_err(y, g[name + "_y"]) < 2e-5 * sc


@pytest.mark.parametrize("name", ["vit1d", "vit3d", "vit2d_aniso"])
def test_vit_on_other_grids(golden, name):
    """1-D / 3-D grids and anisotropic patches: torch rearrangement around the compiled token network."""
    from azula_amd.nn import ViT

    g = golden("g13_spatial")
    vit = ViT(**g.meta[name + "_cfg"])
    vit.load_state_dict(synth.synth_state_dict({n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}, 42))
    y = vit.cuda().eval()(g[name + "_x"].cuda(), g["mod"].cuda())
    sc = max(1.0, g[name + "_y"].abs().max().item())
    print(name, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
    assert y.shape == g[name + "_y"].shape and max_err(y, g[name + "_y"]) < 2e-5 * sc


def test_unet_anisotropic_kernel(golden):
    """kernel_size = (3, 5) (azula/nn/unet.py:165-173): the filters sit centred in 5 x 5 ones."""
    from azula_amd.nn import UNet

    g = golden("g13_spatial")
    net = UNet(**g.meta["unet_aniso_cfg"], kernel_size=(3, 5))
    sh = {n: tuple(v) for n, v in g.meta["unet_aniso_shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == sh
    net.load_state_dict(synth.synth_state_dict(sh, 43))
    y = net.cuda().eval()(g["unet_aniso_x"].cuda(), g["mod"].cuda())
    sc = max(1.0, g["unet_aniso_y"].abs().max().item())
    print("UNet kernel (3, 5) max|d|", max_err(y, g["unet_aniso_y"]), "scale", sc)
    assert max_err(y, g["unet_aniso_y"]) < 2e-5 * sc


@pytest.mark.parametrize("name", ["s4", "s4_odd", "s1", "s8"])
def test_unet_other_strides(golden, name):
    """stride 1 / 4 / 8 (azula/nn/unet.py:159-186): stride-s downsampling convolutions, nearest x s upsampling as a right
    shift of the merge convolution's gather (AzConvArgs.up1 = log2 s), narrow before the concatenation on odd sizes."""
    from azula_amd.nn import UNet

    g = golden("g15_strides")
    net = UNet(**g.meta[name + "_cfg"])
    sh = {n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == sh
    net.load_state_dict(synth.synth_state_dict(sh, 51))
    x = g[name + "_x"]
    y = net.cuda().eval()(x.cuda(), g["mod"][: x.shape[0]].cuda())
    sc = max(1.0, g[name + "_y"].abs().max().item())
    print(name, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
    assert y.shape == g[name + "_y"].shape and max_err(y, g[name + "_y"]) < 2e-5 * sc


@pytest.mark.parametrize("name", ["s3", "s3_odd", "s5", "s23", "s6_periodic"])
def test_unet_strides_that_are_not_powers_of_two(golden, name):
    """stride 3 / 5 / (2, 3) / 6 (azula/nn/unet.py:155-186, 250-254): the stride-s convolution on the direct kernel, the nearest
    x s upsampling + narrow as az_upsample_nearest_f32 in front of the merge convolution (ATen's fp32 source index)."""
    from azula_amd.nn import UNet

    g = golden("g18_odd_strides")
    cfg = dict(g.meta[name + "_cfg"])
    periodic = cfg.pop("periodic")
    if isinstance(cfg["stride"], list):
        cfg["stride"] = tuple(cfg["stride"])
    net = UNet(**cfg, periodic=periodic)
    sh = {n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == sh
    net.load_state_dict(synth.synth_state_dict(sh, 61))
    x = g[name + "_x"]
    net = net.cuda().eval()
    y = net(x.cuda(), g["mod"][: x.shape[0]].cuda())
    ops = [n for _, _, n in next(iter(net._plans.values())).tape.ops]
    assert "az_upsample_nearest_f32" in ops
    sc = max(1.0, g[name + "_y"].abs().max().item())
    print(name, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
    assert y.shape == g[name + "_y"].shape and max_err(y, g[name + "_y"]) < 2e-5 * sc


@pytest.mark.parametrize("name", ["v_even", "v_odd", "v_periodic", "v_layer"])
def test_unet_on_volumes(golden, name):
    """``spatial = 3`` (azula/nn/unet.py:119-259 with Conv3d): every 3-D convolution as depth taps of the 2-D kernels
    accumulating in place (azula_amd/nn/unet3d.py); zero and circular padding, odd sizes (narrow before the concat)."""
    from azula_amd.nn import UNet

    g = golden("g16_unet3d")
    cfg = dict(g.meta[name + "_cfg"])
    periodic = cfg.pop("periodic")
    net = UNet(**cfg, spatial=3, periodic=periodic)
    sh = {n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == sh
    net.load_state_dict(synth.synth_state_dict(sh, 61))
    x = g[name + "_x"]
    y = net.cuda().eval()(x.cuda(), g["mod"][: x.shape[0]].cuda())
    sc = max(1.0, g[name + "_y"].abs().max().item())
    print(name, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
    assert y.shape == g[name + "_y"].shape and max_err(y, g[name + "_y"]) < 2e-5 * sc


@pytest.mark.parametrize("name", ["s3", "s3_ragged", "s312", "s135_periodic"])
def test_unet_on_volumes_with_strides_that_are_not_powers_of_two(golden, name):
    """``UNet(spatial=3, stride=3 | (3, 1, 2) | (1, 3, 5))`` (azula/nn/unet.py:159-186,250-255): the strided convolutions keep every
    s-th plane of a depth-tap launch, the decoder's ``Upsample(nearest) + narrow`` is an in-plane pass + a gather of whole planes
    with ATen's source index (G21: reference outputs; sizes that do not divide, zero and circular padding)."""
    from azula_amd.nn import UNet

    g = golden("g21_unet3d_odd_strides")
    cfg = dict(g.meta[name + "_cfg"])
    periodic = cfg.pop("periodic")
    if isinstance(cfg["stride"], list):
        cfg["stride"] = tuple(cfg["stride"])
    net = UNet(**cfg, spatial=3, periodic=periodic)
    sh = {n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == sh
    net.load_state_dict(synth.synth_state_dict(sh, 62))
    x = g[name + "_x"]
    y = net.cuda().eval()(x.cuda(), g["mod"][: x.shape[0]].cuda())
    sc = max(1.0, g[name + "_y"].abs().max().item())
    print(name, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
    assert y.shape == g[name + "_y"].shape and max_err(y, g[name + "_y"]) < 2e-5 * sc


@pytest.mark.parametrize("periodic", [False, True])
def test_unet_block_on_a_volume(periodic):
    """Standalone ``UNetBlock(spatial=3).forward`` (one-block plan on the volume form) against the oracle's block
    (pinned to the reference for volumes by G16)."""
    from azula_amd.nn.unet import UNetBlock
    from oracle import nets

    torch.manual_seed(5)
    kw = dict(padding_mode="circular") if periodic else {}
    blk = UNetBlock(12, mod_features=8, norm="group", groups=4, spatial=3, kernel_size=(3, 3, 3), **kw)
    for p in blk.parameters():
        p.data.normal_(0, 0.3)
    x, mod = torch.randn(2, 12, 3, 6, 5), torch.randn(2, 8)
    sd = {"b." + k: v.detach().clone() for k, v in blk.state_dict().items()}
    ref = nets.unet_block(sd, "b", x, mod, "group", 4, periodic)
    y = blk.cuda().eval()(x.cuda(), mod.cuda())
    sc = max(1.0, ref.abs().max().item())
    print("UNetBlock 3-D periodic", periodic, "max|d|", max_err(y, ref), "scale", sc)
    assert y.shape == ref.shape and max_err(y, ref) < 2e-5 * sc


def test_sampling_a_volume_with_a_3d_unet():
    """KarrasDenoiser(TimeModulated(UNet(spatial=3))) + DDIM on a (B, C, D, H, W) latent, captured like the image loop (the
    transition kernel sees the volume as an image of D H x W pixels), against the oracle's loop around its 3-D network."""
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import UNet
    from azula_amd.nn.wrappers import TimeModulated
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler
    from oracle import nets, sampling

    torch.manual_seed(8)
    cfg = dict(in_channels=2, out_channels=2, hid_channels=(8, 16), hid_blocks=(1, 1), norm="group", groups=4, mod_features=16)
    bb = TimeModulated(UNet(**cfg, spatial=3), 16, name="unet")
    for p in bb.parameters():
        if p.ndim > 1 and not torch.any(p != 0):
            p.data.normal_(0, 0.1)
    den = KarrasDenoiser(bb, VPSchedule()).cuda().eval()
    sd = {k: v.detach().cpu().clone() for k, v in bb.state_dict().items()}
    x1 = torch.randn(2, 2, 4, 6, 6)
    omean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, cfg, a, c), x, t)  # noqa: E731
    ref = sampling.sample(omean, x1, steps=4, eta=0.0)
    smp = DDIMSampler(den, steps=4, silent=True)
    x0 = smp(x1.cuda())
    assert next(iter(smp._fused_cache.values())).graph is not None, "the volume loop must be captured"
    sc = max(1.0, ref.abs().max().item())
    print("3-D DDIM-4 max|d|", max_err(x0, ref), "scale", sc)
    assert x0.shape == ref.shape and max_err(x0, ref) < 2e-5 * sc


@pytest.mark.parametrize("case", ["k5_cond", "stride4", "shared_mod", "anisotropic_kernel"])
def test_unet_on_volumes_more_shapes(case):
    """spatial = 3 beyond G16's configurations, against the oracle (pinned to the reference for volumes by G16): 5^3
    kernels with a condition volume, stride 4 on an odd volume, a single modulation vector, a (1, 3, 5) kernel."""
    from azula_amd.nn import UNet
    from oracle import nets

    torch.manual_seed({"k5_cond": 1, "stride4": 2, "shared_mod": 3, "anisotropic_kernel": 4}[case])
    cfg = dict(in_channels=2, out_channels=2, hid_channels=(8, 12), hid_blocks=(1, 1), norm="group", groups=4, mod_features=8)
    kw, shape, cond, mod = {}, (2, 2, 5, 6, 7), None, torch.randn(2, 8)
    if case == "k5_cond":
        kw, cond = dict(kernel_size=5, cond_channels=1), torch.randn(2, 1, 5, 6, 7)
    elif case == "stride4":
        kw, shape = dict(stride=4), (1, 2, 9, 10, 6)
        mod = mod[:1]
    elif case == "shared_mod":
        mod = torch.randn(8)
    else:
        kw = dict(kernel_size=(1, 3, 5))
    net = UNet(**cfg, **kw, spatial=3, periodic=case == "shared_mod")
    for p in net.parameters():
        p.data.normal_(0, 0.2)
    x = torch.randn(*shape)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ocfg = dict(cfg, periodic=case == "shared_mod", stride=kw.get("stride", 2))
    ref = nets.unet_forward(sd, ocfg, x if cond is None else torch.cat((x, cond), 1), mod)
    y = net.cuda().eval()(x.cuda(), mod.cuda(), cond=cond.cuda() if cond is not None else None)
    sc = max(1.0, ref.abs().max().item())
    print(case, "max|d|", max_err(y, ref), "scale", sc)
    assert y.shape == ref.shape and max_err(y, ref) < 3e-5 * sc


@pytest.mark.parametrize("name", ["i21", "i14", "i42_periodic", "v122", "v214"])
def test_unet_one_stride_per_axis(golden, name):
    """A stride SEQUENCE (azula/nn/unet.py:159-186): the downsampling convolutions and the nearest upsampling of the merge
    convolution's gather take one (power-of-two) factor per axis (AzConvArgs.aniso / stride_w / up1_w; depth by plane
    index on volumes)."""
    from azula_amd.nn import UNet

    g = golden("g17_anisotropic_strides")
    cfg = dict(g.meta[name + "_cfg"])
    cfg["stride"] = tuple(cfg["stride"])
    net = UNet(**cfg)
    sh = {n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == sh
    net.load_state_dict(synth.synth_state_dict(sh, 71))
    x = g[name + "_x"]
    y = net.cuda().eval()(x.cuda(), g["mod"][: x.shape[0]].cuda())
    sc = max(1.0, g[name + "_y"].abs().max().item())
    print(name, "max|d|", max_err(y, g[name + "_y"]), "scale", sc)
    assert y.shape == g[name + "_y"].shape and max_err(y, g[name + "_y"]) < 2e-5 * sc


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
def test_unet_on_volumes_half_precision(golden, half):
    """A volume network cast to half precision runs its plane convolutions on the bf16 / f16 MFMA kernel (fp32 accumulate,
    in-place accumulation over the depth taps in fp32): against the fp32 reference output at the half-precision bar."""
    from azula_amd.nn import UNet

    g = golden("g16_unet3d")
    cfg = dict(g.meta["v_even_cfg"])
    periodic = cfg.pop("periodic")
    net = UNet(**cfg, spatial=3, periodic=periodic)
    net.load_state_dict(synth.synth_state_dict({n: tuple(v) for n, v in g.meta["v_even_shapes"].items()}, 61))
    x = g["v_even_x"]
    y = net.cuda().eval().to(half)(x.cuda().to(half), g["mod"][: x.shape[0]].cuda().to(half))
    ref = g["v_even_y"]
    sc = max(1.0, ref.abs().max().item())
    err = max_err(y.float(), ref)
    print(half, "max|d|", err, "scale", sc)
    assert y.dtype == half and err < (4e-2 if half == torch.bfloat16 else 5e-3) * sc  # measured 2.1e-2 / 2.5e-3 on scale 2.8
