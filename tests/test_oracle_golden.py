r"""The oracle (CPU restatement) against the committed golden vectors.

The vectors were produced by the REFERENCE itself (``oracle/make_golden.py``, which also
asserted bit-equality oracle == reference in the build container).  Here the oracle is re-run
on whatever host executes the tests; elementwise results must agree to the last bit or two
(vector-ISA dependent libm), network results to fp32 round-off.
"""

import torch

from conftest import max_err
from oracle import nets, sampling, synth

torch.set_grad_enabled(False)
TIGHT = dict(rtol=2e-7, atol=1e-9)


def test_g1_schedule_tables(golden):
    g = golden("g1_schedule")
    for case in g.meta["cases"]:
        tag = case["tag"]
        ts = sampling.timesteps(steps=case["steps"])
        assert torch.equal(ts, g[tag + "_t"])
        al, si = zip(*(sampling.vp_schedule(t, case["alpha_min"], case["sigma_min"]) for t in ts.unbind()))
        torch.testing.assert_close(torch.stack(al), g[tag + "_alpha"], **TIGHT)
        torch.testing.assert_close(torch.stack(si), g[tag + "_sigma"], **TIGHT)
        assert ts[0] == 1 and ts[-1] == 0  # G7: endpoints
        assert g[tag + "_alpha"][-1] == 1.0  # alpha_0 == 1 (reference tests/test_noise.py)


def test_g2_preconditioning(golden):
    g = golden("g2_precond")
    ts = g["t"]
    rows = []
    for t in ts.unbind():
        a, s = sampling.vp_schedule(t)
        rows.append(torch.stack(sampling.karras_coefficients(a, s)))
    torch.testing.assert_close(torch.stack(rows), g["karras"], **TIGHT)
    sig = sampling.adm_sigmas("linear", 1000)
    torch.testing.assert_close(sig, g["adm_sigmas"], **TIGHT)
    torch.testing.assert_close(sampling.adm_sigmas("cosine", 1000), g["adm_sigmas_cosine"], **TIGHT)
    idx = []
    for t in ts[:-1].unbind():
        a, s = sampling.vp_schedule(t, 1e-2, 1e-2)
        idx.append(sampling.adm_coefficients(a, s, g["adm_sigmas"])[3][0])
    assert torch.equal(torch.stack(idx), g["adm_idx"])
    assert g["adm_idx"][0] == 954 and g["adm_idx"][-1] == 11  # DDIM-64 indices 954, 939, ..., 11


def test_g3_single_transition(golden):
    g = golden("g3_transition")
    for case in g.meta["cases"]:
        t, s = torch.tensor(case["t"]), torch.tensor(case["s"])
        mean = sampling.karras_mean(lambda x, c: g["F"], g["x_t"], t)
        torch.testing.assert_close(mean, g[case["tag"] + "_mean"], rtol=1e-6, atol=1e-6)
        a_t, s_t = sampling.vp_schedule(t)
        a_s, s_s = sampling.vp_schedule(s)
        x_s = sampling.transition(g["x_t"], g[case["tag"] + "_mean"], g["eps"], a_t, s_t, a_s, s_s, case["eta"])
        torch.testing.assert_close(x_s, g[case["tag"]], rtol=1e-6, atol=1e-6)


def _toy(sd):
    def f(x, c_time, **_):
        h = torch.nn.functional.linear(x, sd["l1.weight"], sd["l1.bias"]) + nets.sine_encoding(c_time, 64)
        return torch.nn.functional.linear(torch.relu(h), sd["l2.weight"], sd["l2.bias"])

    return f


def test_g4_toy_loop(golden):
    g = golden("g4_toy_loop")
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
    mean = lambda x, t: sampling.karras_mean(_toy(sd), x, t)  # noqa: E731
    x0 = sampling.sample(mean, g["x1"], steps=64, eta=0.0)
    assert max_err(x0, g["ddim64"]) < 1e-5
    x0 = sampling.sample(mean, g["x1"], steps=64, eta=None, eps_list=list(g["ddpm64_eps"]))
    assert max_err(x0, g["ddpm64"]) < 1e-5
    # G7: DDIM(eta=1) == DDPM given identical noise
    x0b = sampling.sample(mean, g["x1"], steps=64, eta=1.0, eps_list=list(g["ddpm64_eps"]))
    assert max_err(x0b, x0) < 1e-6
    # same generator stream as the reference run
    torch.manual_seed(g.meta["loop_seed"])
    x0 = sampling.sample(mean, g["x1"], steps=1000, eta=None)
    assert max_err(x0, g["ddpm1000"]) < 1e-4


def test_g5_unets(golden):
    for name in ("unet_group", "unet_layer_odd", "unet_rms_nomod"):
        g = golden("g5_" + name)
        cfg = g.meta["cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
        tap = {}
        y = nets.unet_forward(sd, cfg, g["x"], g["modB"] if "modB" in g else None, tap=tap)
        assert max_err(y, g["y_modB"]) < 1e-5, name
        for k, v in tap.items():
            assert max_err(v, g["tap_" + k]) < 1e-5, (name, k)
        if "mod1" in g:
            assert max_err(nets.unet_forward(sd, cfg, g["x"], g["mod1"]), g["y_mod1"]) < 1e-5


def test_g5_vit(golden):
    g = golden("g5_vit")
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
    y = nets.vit_forward(sd, g.meta["cfg"], g["x"], g["modB"])
    assert max_err(y, g["y_modB"]) < 1e-5
    assert max_err(nets.vit_forward(sd, g.meta["cfg"], g["x"], g["mod1"]), g["y_mod1"]) < 1e-5


def test_g5_adm(golden):
    for name in ("adm_uncond", "adm_cond_neworder"):
        g = golden("g5_" + name)
        cfg = g.meta["cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
        y = g["y"] if "y" in g else None
        out = nets.adm_unet_forward(sd, cfg, g["x"], g["idx"], y)
        assert max_err(out, g["out"]) < 2e-5, name
        sig = sampling.adm_sigmas(cfg["discrete_schedule"], cfg["discrete_steps"])
        bb = lambda a, i, y=None: nets.adm_unet_forward(sd, cfg, a, i, y)  # noqa: E731
        mean, var = sampling.adm_posterior(bb, g["x"], torch.tensor(0.7), sig, label=y)
        assert max_err(mean, g["mean_t07"]) < 2e-5 and max_err(var, g["var_t07"]) < 2e-5


def test_g6_unet_loop(golden):
    g = golden("g6_unet_loop")
    cfg = g.meta["cfg"]
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
    mean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, cfg, a, c), x, t)  # noqa: E731
    assert max_err(mean(g["x1"], torch.tensor(0.5)), g["mean_t05"]) < 1e-5
    x0 = sampling.sample(mean, g["x1"], steps=8, eta=None, eps_list=list(g["ddpm8_eps"]))
    assert max_err(x0, g["ddpm8"]) < 1e-4


MULTISTEP = ("zAB", "vAB", "zEAB", "xEAB", "REAB")


def test_g9_multistep_weights(golden):
    """The fp64 Vandermonde solves of the AB family, rounded to fp32, against the reference's."""
    g = golden("g9_multistep_weights")
    alpha, sigma = g["alpha"], g["sigma"]
    for kind in MULTISTEP:
        u = sigma / alpha if kind == "zAB" else sigma / (alpha + sigma) if kind == "vAB" else sigma.log() - alpha.log()
        for order in (1, 2, 3, 4):
            want = g[f"{kind}_w{order}"]
            for i in range(g.meta["steps"]):
                c = sampling.multistep_weights(kind, u, i, order)
                torch.testing.assert_close(c, want[i, : len(c)], rtol=2e-5, atol=1e-7, msg=f"{kind} {order} {i}")
    # first-order rules collapse to closed forms: AB -> u_s - u_t, zEAB -> e^{u_s} - e^{u_t}
    u = (sigma / alpha).double()
    assert abs(float(sampling.multistep_weights("zAB", u, 5, 1)) - float(u[6] - u[5])) < 1e-12 * float(u[5])
    u = (sigma.log() - alpha.log()).double()
    assert abs(float(sampling.multistep_weights("zEAB", u, 5, 1)) - float(u[6].exp() - u[5].exp())) < 1e-9


def test_g9_toy_multistep_loops(golden):
    g = golden("g9_toy_multistep")
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
    mean = lambda x, t: sampling.karras_mean(_toy(sd), x, t)  # noqa: E731
    for kind in MULTISTEP:
        for order in (1, 2, 3):
            x0 = sampling.sample_multistep(mean, g["x1"], kind, order=order, steps=g.meta["steps"])
            assert max_err(x0, g[f"{kind}_o{order}"]) < 2e-5, (kind, order)
    x0 = sampling.sample_pc(mean, g["x1"], steps=g.meta["steps"], eps_list=list(g["pc_eps"]), **g.meta["pc"])
    assert max_err(x0, g["pc"]) < 1e-5
    # order-1 zAB is the Euler step and order-1 zEAB/xEAB the DDIM step: same ODE, same answer at fp32 level
    assert max_err(g["zAB_o1"], sampling.sample_euler(mean, g["x1"], steps=g.meta["steps"])) < 1e-4
    assert max_err(g["xEAB_o1"], sampling.sample(mean, g["x1"], steps=g.meta["steps"], eta=0.0)) < 1e-4


def test_g10_jit(golden):
    """JiT backbone, JITDenoiser (label / null class), DDIM and CFG loops on the rectified schedule."""
    for name in ("g10_jit_ctx", "g10_jit_noctx_hd32", "g10_jit_hd80", "g24_jit_hd48"):
        g = golden(name)
        cfg = g.meta["cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
        bb = lambda a, c, l: nets.jit_forward(sd, cfg, a, c, l)  # noqa: E731
        y = g["y"].long()
        sc = g["out"].abs().max().item()
        assert max_err(bb(g["x"], g["t"], y), g["out"]) < 2e-5 * sc, name
        assert max_err(bb(g["x"], g["t"][:1], y), g["out_shared"]) < 2e-5 * sc, name
        mean = lambda a, tt, label=None: sampling.jit_mean(bb, a, tt, label, cfg["num_classes"])  # noqa: E731
        assert max_err(mean(g["x"], torch.tensor(0.4), y), g["mean_t04"]) < 2e-5 * sc
        assert max_err(mean(g["x"], torch.tensor(0.7)), g["mean_null"]) < 2e-5 * sc
        x0 = sampling.sample(mean, g["x1"], schedule=sampling.rectified_schedule, steps=8, eta=0.0, label=y)
        assert max_err(x0, g["ddim8"]) < 1e-4 * max(1.0, g["ddim8"].abs().max().item()), name
        cm = lambda a, tt: sampling.cfg_mean(mean, a, tt, {"label": y}, {}, g.meta["guidance"])  # noqa: E731
        x0 = sampling.sample(cm, g["x1"], schedule=sampling.rectified_schedule, steps=6, eta=0.0)
        assert max_err(x0, g["cfg_ddim6"]) < 2e-4 * max(1.0, g["cfg_ddim6"].abs().max().item()), name


def test_g12_blocks_mask_cond_periodic(golden):
    """Standalone block forwards, the attention mask, ViT cond / unpatch_size and the periodic UNet."""
    g = golden("g12_blocks_mask_cond_periodic")
    sh = lambda k: {n: tuple(v) for n, v in g.meta[k].items()}  # noqa: E731
    mod = g["mod"]
    msd = {("." + k): v for k, v in synth.synth_state_dict(sh("msa_shapes"), 31).items()}
    causal, bmask = g["msa_causal"].bool(), g["msa_bmask"].bool()
    assert max_err(nets.msa_forward(msd, "", g["msa_x"], 4, pos=g["msa_pos"], mask=causal), g["msa_y_causal"]) < 1e-5
    assert max_err(nets.msa_forward(msd, "", g["msa_x"], 4, pos=g["msa_pos"], mask=bmask), g["msa_y_bmask"]) < 1e-5
    assert max_err(nets.msa_forward(msd, "", g["msa_x"], 4, pos=g["msa_pos"]), g["msa_y_nomask"]) < 1e-5
    bsd = {("b." + k): v for k, v in synth.synth_state_dict(sh("dit_shapes"), 32).items()}
    y = nets.dit_block(bsd, "b", g["msa_x"], g["dit_mod"], 4, pos=g["msa_pos"], act="swiglu", mask=causal)
    assert max_err(y, g["dit_y"]) < 1e-5
    usd = {("u." + k): v for k, v in synth.synth_state_dict(sh("ublock_shapes"), 33).items()}
    assert max_err(nets.unet_block(usd, "u", g["ublock_x"], mod, "group", 4), g["ublock_y"]) < 1e-5
    vsd = synth.synth_state_dict(sh("vit_shapes"), 35)
    assert max_err(nets.vit_forward(vsd, g.meta["vit_cfg"], g["vit_x"], mod, cond=g["vit_cond"]), g["vit_y"]) < 1e-5
    v3 = synth.synth_state_dict(sh("vit3_shapes"), 36)
    assert max_err(nets.vit_forward(v3, g.meta["vit3_cfg"], g["vit3_x"], mod[0]), g["vit3_y"]) < 1e-5
    psd = synth.synth_state_dict(sh("punet_shapes"), 37)
    pcfg = dict(g.meta["punet_cfg"], periodic=True)
    for name in ("punet_a", "punet_b", "punet_c"):
        x = g[name + "_x"]
        assert max_err(nets.unet_forward(psd, pcfg, x, mod[: x.shape[0]]), g[name + "_y"]) < 1e-5


def test_g13_other_spatial_dimensions(golden):
    """UNet on 1-D signals (also periodic) and ViT on 1-D / 3-D grids / anisotropic patches."""
    g = golden("g13_spatial")
    mod = g["mod"]
    for name in ("unet1d", "unet1d_odd", "unet1d_periodic"):
        sd = synth.synth_state_dict({n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}, 41)
        assert max_err(nets.unet_forward(sd, g.meta[name + "_cfg"], g[name + "_x"], mod), g[name + "_y"]) < 1e-5
    for name in ("vit1d", "vit3d", "vit2d_aniso"):
        sd = synth.synth_state_dict({n: tuple(v) for n, v in g.meta[name + "_shapes"].items()}, 42)
        assert max_err(nets.vit_forward(sd, g.meta[name + "_cfg"], g[name + "_x"], mod), g[name + "_y"]) < 1e-5


ADM_FIXTURES = [("g14_" + n) for n in ("adm_plain_conv", "adm_plain_pool", "adm_film_noupdown")] + [
    ("g22_" + n) for n in ("adm_1d_film_updown", "adm_1d_plain_conv", "adm_1d_plain_pool")
] + [("g23_" + n) for n in ("adm_3d_film_updown", "adm_3d_plain_conv", "adm_3d_plain_pool")] + [
    ("g24_" + n) for n in ("adm_hd24_legacy", "adm_hd48_hd96_neworder")  # head sizes 24 / 48 / 96 (--only-g24)
]


def test_g14_g22_g23_adm_offcard_1d_and_3d(golden):
    """guided-diffusion's default wiring (h + emb, Downsample / Upsample layers with and without conv_resample; G14),
    ``dims=1`` signals (G22) and ``dims=3`` volumes (G23): the oracle against the reference's outputs
    (oracle/make_golden.py --only-g14 / --only-g22 / --only-g23)."""
    for fixture in ADM_FIXTURES:
        g = golden(fixture)
        cfg = g.meta["cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
        y = g["y"] if "y" in g else None
        out = nets.adm_unet_forward(sd, cfg, g["x"], g["idx"], y)
        torch.testing.assert_close(out, g["out"], **TIGHT)
        sig = sampling.adm_sigmas(cfg["discrete_schedule"], cfg["discrete_steps"])
        bb = lambda a, i, y=None: nets.adm_unet_forward(sd, cfg, a, i, y)  # noqa: E731
        omean = lambda xx, t, label=None: sampling.adm_posterior(bb, xx, t, sig, label=label)[0]  # noqa: E731
        kw = {"label": y} if y is not None else {}
        x0 = sampling.sample(omean, g["x1"], schedule=lambda t: sampling.vp_schedule(t, 1e-2, 1e-2), steps=8, eta=0.0, **kw)
        torch.testing.assert_close(x0, g["ddim8"], **TIGHT)


def test_g24_vit_head_sizes(golden):
    """ViT with head sizes the gfx950 attention kernels are not instantiated for (48 with RoPE + q/k norm, 96, 24): the oracle
    against the reference's outputs (oracle/make_golden.py --only-g24)."""
    for name in ("vit_hd48_rope", "vit_hd96_noqknorm", "vit_hd24"):
        g = golden("g24_" + name)
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["shapes"].items()}, g.meta["weight_seed"])
        assert max_err(nets.vit_forward(sd, g.meta["cfg"], g["x"], g["modB"]), g["y_modB"]) < 1e-5, name


def test_g15_strides(golden):
    g = golden("g15_strides")
    for name in ("s4", "s4_odd", "s1", "s8"):
        cfg = g.meta[name + "_cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta[name + "_shapes"].items()}, 51)
        x = g[name + "_x"]
        y = nets.unet_forward(sd, cfg, x, g["mod"][: x.shape[0]])
        torch.testing.assert_close(y, g[name + "_y"], **TIGHT)


def test_g19_adm_fractional_timesteps(golden):
    g = golden("g19_adm_fractional")
    for name in ("adm_uncond", "adm_cond_neworder"):
        cfg = g.meta[name + "_cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta[name + "_shapes"].items()}, 9)
        y = g[name + "_y"] if name + "_y" in g else None
        torch.testing.assert_close(nets.adm_unet_forward(sd, cfg, g[name + "_x"], g[name + "_t"], y), g[name + "_out"], **TIGHT)
        torch.testing.assert_close(nets.adm_unet_forward(sd, cfg, g[name + "_x"], torch.tensor([417.75]), y), g[name + "_out_shared"], **TIGHT)


def test_g18_odd_strides(golden):
    g = golden("g18_odd_strides")
    for name in ("s3", "s3_odd", "s5", "s23", "s6_periodic"):
        cfg = g.meta[name + "_cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta[name + "_shapes"].items()}, 61)
        x = g[name + "_x"]
        y = nets.unet_forward(sd, cfg, x, g["mod"][: x.shape[0]])
        torch.testing.assert_close(y, g[name + "_y"], **TIGHT)


def test_g16_unet3d(golden):
    g = golden("g16_unet3d")
    for name in ("v_even", "v_odd", "v_periodic", "v_layer"):
        cfg = g.meta[name + "_cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta[name + "_shapes"].items()}, 61)
        x = g[name + "_x"]
        y = nets.unet_forward(sd, cfg, x, g["mod"][: x.shape[0]])
        torch.testing.assert_close(y, g[name + "_y"], **TIGHT)


def test_g21_unet3d_odd_strides(golden):
    g = golden("g21_unet3d_odd_strides")
    for name in ("s3", "s3_ragged", "s312", "s135_periodic"):
        cfg = g.meta[name + "_cfg"]
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta[name + "_shapes"].items()}, 62)
        x = g[name + "_x"]
        y = nets.unet_forward(sd, cfg, x, g["mod"][: x.shape[0]])
        torch.testing.assert_close(y, g[name + "_y"], **TIGHT)


def test_g17_anisotropic_strides(golden):
    g = golden("g17_anisotropic_strides")
    for name in ("i21", "i14", "i42_periodic", "v122", "v214"):
        cfg = dict(g.meta[name + "_cfg"])
        cfg.pop("spatial")
        sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta[name + "_shapes"].items()}, 71)
        x = g[name + "_x"]
        y = nets.unet_forward(sd, cfg, x, g["mod"][: x.shape[0]])
        torch.testing.assert_close(y, g[name + "_y"], **TIGHT)


def _g20_oracle(g):
    cfg = g.meta["unet_cfg"]
    sd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["unet_shapes"].items()}, g.meta["unet_weight_seed"])
    bb = lambda a, c, **_: nets.time_wrapped_unet(sd, cfg, a, c)  # noqa: E731
    scheds = {"cosine": sampling.cosine_schedule, "rectified": sampling.rectified_schedule}
    dens = {"karras": sampling.karras_mean, "simple": sampling.simple_mean}
    return bb, scheds, dens


def test_g20_usages(golden):
    """Per-sample times, start / stop, cosine / rectified schedules x Karras / Simple denoisers x sampler families, ADM with
    per-sample times and a tensor-valued guidance strength: the oracle against the reference's outputs."""
    g = golden("g20_usages")
    bb, scheds, dens = _g20_oracle(g)
    x = g["x"]
    assert max_err(sampling.karras_mean(bb, x, g["t_per_sample"]), g["mean_per_sample_t"]) < 1e-5
    om = lambda xx, t: sampling.karras_mean(bb, xx, t)  # noqa: E731
    assert max_err(sampling.sample(om, x, steps=5, eta=0.0, start=0.8, stop=0.1), g["ddim5_08_01"]) < 2e-5
    for sname, sora in scheds.items():
        for dname, dora in dens.items():
            key = f"{sname}_{dname}"
            m = lambda xx, t, dora=dora, sora=sora: dora(bb, xx, t, schedule=sora)  # noqa: E731
            x1 = g[key + "_x1"]
            sc = lambda k: 2e-5 * max(1.0, g[k].abs().max().item())  # noqa: E731
            assert max_err(sampling.sample(m, x1, schedule=sora, steps=4, eta=0.0), g[key + "_ddim"]) < sc(key + "_ddim")
            assert max_err(sampling.sample_euler(m, x1, schedule=sora, steps=4), g[key + "_euler"]) < sc(key + "_euler")
            assert max_err(sampling.sample_euler(m, x1, schedule=sora, steps=4, heun=True), g[key + "_heun"]) < sc(key + "_heun")
            assert max_err(sampling.sample_multistep(m, x1, "zEAB", order=2, schedule=sora, steps=4), g[key + "_zeab"]) < sc(key + "_zeab")
    acfg = g.meta["adm_cfg"]
    asd = synth.synth_state_dict({k: tuple(v) for k, v in g.meta["adm_shapes"].items()}, g.meta["adm_weight_seed"])
    sig = sampling.adm_sigmas(acfg["discrete_schedule"], acfg["discrete_steps"])
    abb = lambda a, i, y=None: nets.adm_unet_forward(asd, acfg, a, i, y)  # noqa: E731
    mean, var = sampling.adm_posterior(abb, g["adm_x"], g["adm_t"], sig, label=g["adm_label"])
    assert max_err(mean, g["adm_mean"]) < 5e-5 and max_err(var, g["adm_var"]) < 5e-5 * max(1.0, g["adm_var"].abs().max().item())
