r"""Per-kernel parity: each C-ABI entry point against the oracle / a plain torch fp32 CPU
reference of the same op, called exactly as the product calls it (ctypes -> HIP)."""

import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import max_err

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def az():
    from azula_amd import _lib

    _lib.lib()
    return _lib


def dev(t):
    return t.to("cuda").contiguous()


def to_nhwc(x, cs=None):
    B, Cc, H, W = x.shape
    cs = cs or (Cc + 3) // 4 * 4
    y = torch.zeros(B, H, W, cs, dtype=x.dtype, device=x.device)
    y[..., :Cc] = x.permute(0, 2, 3, 1)
    return y.contiguous()


def from_nhwc(y, Cc):
    return y[..., :Cc].permute(0, 3, 1, 2).contiguous()


def coef_row(**kw):
    from azula_amd._lib import COEF_FIELDS, COEF_WORDS

    row = torch.zeros(COEF_WORDS, dtype=torch.float32)
    row[COEF_FIELDS.index("clip_lo")] = -math.inf
    row[COEF_FIELDS.index("clip_hi")] = math.inf
    for k, v in kw.items():
        if k in ("time_index", "step"):
            row.view(torch.int32)[COEF_FIELDS.index(k)] = int(v)
        else:
            row[COEF_FIELDS.index(k)] = float(v)
    return row


def ref_transition(x, Fp, Fn, eps, k):
    """torch-CPU op sequence of denoise.py:322 (+clip, +cfg) and sample.py:257-259."""
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
    mean = f32(k["c_skip"]) * x + f32(k["c_out"]) * Fp
    mean = torch.clip(mean, min=k.get("clip_lo", -math.inf), max=k.get("clip_hi", math.inf))
    if Fn is not None:
        mn = f32(k["c_skip"]) * x + f32(k["c_out"]) * Fn
        mn = torch.clip(mn, min=k.get("clip_lo", -math.inf), max=k.get("clip_hi", math.inf))
        mean = mean + f32(k["guidance"]) * (mean - mn)
    xs = f32(k["alpha_s"]) * mean
    xs = xs + f32(k["k_x"]) * (x - f32(k["alpha_t"]) * mean)
    if eps is not None:
        xs = xs + f32(k["k_eps"]) * eps
    return mean, xs, f32(k.get("c_in_next", 0.0)) * xs


COEFS = dict(c_skip=0.37, c_out=-1.9, alpha_t=0.61, alpha_s=0.83, k_x=0.71, k_eps=0.29, c_in_next=1.3, guidance=2.0)


@pytest.mark.parametrize("n", [4096, 1027, 3])
@pytest.mark.parametrize("cfg", [False, True])
@pytest.mark.parametrize("use_eps", [False, True])
def test_transition_flat_bit_exact(az, n, cfg, use_eps):
    g = torch.Generator().manual_seed(n)
    x, Fp, Fn, eps = (torch.randn(n, generator=g) for _ in range(4))
    k = dict(COEFS, clip_lo=-1.0, clip_hi=1.0) if cfg else dict(COEFS)
    mean, xs, xin = ref_transition(x, Fp, Fn if cfg else None, eps if use_eps else None, k)
    row = dev(coef_row(**k))
    dx, dF, dFn, de = dev(x), dev(Fp), dev(Fn), dev(eps)
    o_xs, o_xin, o_mean = (torch.empty(n, device="cuda") for _ in range(3))
    a = az.AzTransitionArgs(
        x_t=dx.data_ptr(), F=dF.data_ptr(), F_neg=dFn.data_ptr() if cfg else None,
        eps=de.data_ptr() if use_eps else None, x_s=o_xs.data_ptr(), xin_next=o_xin.data_ptr(),
        mean_out=o_mean.data_ptr(), batch=1, channels=1, inner=n, f_channels=1, coef=row.data_ptr(),
    )
    az.call("az_transition_f32", C.byref(a), az.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(o_mean.cpu(), mean)
    assert torch.equal(o_xs.cpu(), xs)
    assert torch.equal(o_xin.cpu(), xin)


def test_transition_flat_in_place_and_golden(az, golden):
    """Against the reference-generated G3 vectors, x_s aliasing x_t."""
    from oracle import sampling

    g = golden("g3_transition")
    for case in g.meta["cases"]:
        t, s = torch.tensor(case["t"]), torch.tensor(case["s"])
        a_t, s_t = sampling.vp_schedule(t)
        a_s, s_s = sampling.vp_schedule(s)
        c_in, c_out, c_skip, _ = sampling.karras_coefficients(a_t, s_t)
        tau = 1 - (a_t / a_s * s_s / s_t) ** 2
        if case["eta"] is not None:
            tau = torch.clip(case["eta"] * tau, min=0, max=1)
        row = dev(coef_row(c_skip=c_skip, c_out=c_out, alpha_t=a_t, alpha_s=a_s, k_x=s_s * torch.sqrt(1 - tau) / s_t,
                           k_eps=s_s * torch.sqrt(tau)))
        x, Fb, eps = dev(g["x_t"]), dev(g["F"]), dev(g["eps"])
        n = x.numel()
        arg = az.AzTransitionArgs(x_t=x.data_ptr(), F=Fb.data_ptr(), eps=eps.data_ptr(), x_s=x.data_ptr(), batch=1,
                                  channels=1, inner=n, f_channels=1, coef=row.data_ptr())
        az.call("az_transition_f32", C.byref(arg), az.stream_ptr())
        # the kernel arithmetic is bit-exact (test above); the host-computed scalars may differ in
        # the last ulp between the CPU that wrote the fixture and this one (vector libm variants)
        torch.testing.assert_close(x.cpu(), g[case["tag"]], rtol=2e-6, atol=2e-6, msg=case["tag"])


@pytest.mark.parametrize("n_hist", [0, 1, 2, 7])
@pytest.mark.parametrize("n", [4, 1001, 65536 + 3])
def test_multistep_bit_exact(az, n, n_hist):
    """az_multistep_f32 (AB-family update, azula/sample.py:519-546): separately rounded fp32 mul/add in
    the documented order, float4 body + scalar tail, x_s aliasing x_t."""
    g = torch.Generator().manual_seed(n + n_hist)
    x, mean = torch.randn(n, generator=g), torch.randn(n, generator=g)
    hist = [torch.randn(n, generator=g) for _ in range(n_hist)]
    coef = torch.randn(4 + n_hist, generator=g)
    pred = coef[0] * x + coef[1] * mean
    acc = coef[2] * x
    for j, h in enumerate(hist):
        acc = acc + coef[4 + j] * h
    want = acc + coef[3] * pred
    dx, dm, dc = dev(x), dev(mean), dev(coef)
    dh = [dev(h) for h in hist]
    dpred = torch.empty(n, device="cuda")
    a = az.AzMultistepArgs(x_s=dx.data_ptr(), pred=dpred.data_ptr(), x_t=dx.data_ptr(), mean=dm.data_ptr(),
                           coef=dc.data_ptr(), count=n, n_hist=n_hist)
    for j, h in enumerate(dh):
        a.hist[j] = h.data_ptr()
    az.call("az_multistep_f32", C.byref(a), az.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(dpred.cpu(), pred)
    assert torch.equal(dx.cpu(), want)


def test_multistep_rejects_bad_arguments(az):
    x = torch.zeros(16, device="cuda")
    c = torch.zeros(12, device="cuda")
    a = az.AzMultistepArgs(x_s=x.data_ptr(), pred=x.data_ptr(), x_t=x.data_ptr(), mean=x.data_ptr(), coef=c.data_ptr(),
                           count=16, n_hist=8)
    with pytest.raises(az.AzulaAmdError):
        az.call("az_multistep_f32", C.byref(a), az.stream_ptr())
    a.n_hist = 1
    a.hist[0] = x.data_ptr()  # history slot aliasing the slot being written
    with pytest.raises(az.AzulaAmdError):
        az.call("az_multistep_f32", C.byref(a), az.stream_ptr())


@pytest.mark.parametrize("f_nhwc,fC", [(0, 3), (0, 6), (1, 4), (1, 8)])
@pytest.mark.parametrize("cfg", [False, True])
def test_transition_image_layouts(az, f_nhwc, fC, cfg):
    B, Cc, H, W = 2, 3, 8, 12
    g = torch.Generator().manual_seed(5)
    x, eps = torch.randn(B, Cc, H, W, generator=g), torch.randn(B, Cc, H, W, generator=g)
    Fp, Fn = torch.randn(B, fC, H, W, generator=g), torch.randn(B, fC, H, W, generator=g)
    k = dict(COEFS, clip_lo=-1.0, clip_hi=1.0)
    mean, xs, xin = ref_transition(x, Fp[:, :Cc], Fn[:, :Cc] if cfg else None, eps, k)
    row = dev(coef_row(**k))
    dx, de = dev(x), dev(eps)
    dF = dev(Fp.permute(0, 2, 3, 1)) if f_nhwc else dev(Fp)
    dFn = dev(Fn.permute(0, 2, 3, 1)) if f_nhwc else dev(Fn)
    o_xs, o_mean = torch.empty_like(dx), torch.empty_like(dx)
    o_xin = torch.full((B, H, W, 8), 7.0, device="cuda")
    a = az.AzTransitionArgs(
        x_t=dx.data_ptr(), F=dF.data_ptr(), F_neg=dFn.data_ptr() if cfg else None, eps=de.data_ptr(),
        x_s=o_xs.data_ptr(), xin_next=o_xin.data_ptr(), mean_out=o_mean.data_ptr(), batch=B, channels=Cc,
        inner=H * W, f_channels=fC, f_nhwc=f_nhwc, nhwc_pad=8, coef=row.data_ptr(),
    )
    az.call("az_transition_f32", C.byref(a), az.stream_ptr())
    assert torch.equal(o_xs.cpu(), xs) and torch.equal(o_mean.cpu(), mean)
    assert torch.equal(from_nhwc(o_xin, Cc).cpu(), xin)
    assert (o_xin[..., Cc:] == 0).all()


def test_step_begin_and_scalar_plumbing(az):
    table = torch.stack([coef_row(c_time=0.1 * i, c_in=i, time_index=900 - i, step=i) for i in range(5)])
    dt = dev(table)
    cur = torch.zeros(16, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ct = torch.zeros(1, device="cuda")
    emb = dev(torch.arange(1000 * 8, dtype=torch.float32).reshape(1000, 8))
    row = torch.zeros(8, device="cuda")
    for i in range(5):
        az.call("az_step_begin", cur.data_ptr(), dt.data_ptr(), cnt.data_ptr(), 5, az.stream_ptr())
        az.call("az_coef_c_time_f32", ct.data_ptr(), cur.data_ptr(), az.stream_ptr())
        az.call("az_gather_step_row_f32", row.data_ptr(), emb.data_ptr(), cur.data_ptr(), 0, 8, 1000, az.stream_ptr())
        assert torch.equal(cur.cpu(), table[i]) and cnt.item() == i + 1
        assert ct.item() == table[i, 3].item()
        assert torch.equal(row.cpu(), emb[900 - i].cpu())
    idx = dev(torch.tensor([3, 999, 0]))
    out = torch.zeros(3, 8, device="cuda")
    az.call("az_gather_rows_f32", out.data_ptr(), emb.data_ptr(), idx.data_ptr(), 3, 8, 1000, az.stream_ptr())
    assert torch.equal(out, emb[idx])


def test_scale_axpby_layout(az):
    g = torch.Generator().manual_seed(1)
    x, z = torch.randn(3, 5, 7, generator=g), torch.randn(3, 5, 7, generator=g)
    a, b = torch.randn(3, generator=g), torch.randn(3, generator=g)
    dx, dz, da, db = dev(x), dev(z), dev(a), dev(b)
    y = torch.empty_like(dx)
    az.call("az_scale_f32", y.data_ptr(), dx.data_ptr(), da.data_ptr(), x.numel(), az.stream_ptr())
    assert torch.equal(y.cpu(), a[0] * x)
    az.call("az_axpby_f32", y.data_ptr(), da.data_ptr(), dx.data_ptr(), db.data_ptr(), dz.data_ptr(), 3, 35, 1, az.stream_ptr())
    assert torch.equal(y.cpu(), a[:, None, None] * x + b[:, None, None] * z)
    az.call("az_axpby_f32", y.data_ptr(), da.data_ptr(), dx.data_ptr(), db.data_ptr(), dz.data_ptr(), 1, 105, 0, az.stream_ptr())
    assert torch.equal(y.cpu(), a[0] * x + b[0] * z)
    # NCHW <-> NHWC with channel padding and scale
    img = torch.randn(2, 5, 6, 7, generator=g)
    di = dev(img)
    nh = torch.full((2, 6, 7, 8), 9.0, device="cuda")
    az.call("az_nchw_to_nhwc_f32", nh.data_ptr(), di.data_ptr(), da.data_ptr(), 2, 5, 42, 8, az.stream_ptr())
    assert torch.equal(from_nhwc(nh, 5).cpu(), a[0] * img) and (nh[..., 5:] == 0).all()
    back = torch.empty_like(di)
    az.call("az_nhwc_to_nchw_f32", back.data_ptr(), nh.data_ptr(), 2, 5, 42, 8, az.stream_ptr())
    assert torch.equal(back.cpu(), a[0] * img)


@pytest.mark.parametrize("M,N,K", [(1, 48, 1), (1, 1024, 1024), (3, 100, 16), (5, 7, 40), (64, 96, 64)])
@pytest.mark.parametrize("in_act,out_act", [(0, 0), (0, 1), (1, 0)])
def test_linear_small(az, M, N, K, in_act, out_act):
    g = torch.Generator().manual_seed(M * N + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    ref = F.linear(F.silu(x) if in_act else x, W, b)
    ref = F.silu(ref) if out_act else ref
    dx, dW, db = dev(x), dev(W), dev(b)
    y = torch.empty(M, N, device="cuda")
    az.call("az_linear_small_f32", y.data_ptr(), N, dx.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, N, K, in_act,
            out_act, az.stream_ptr())
    assert max_err(y, ref) < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,in_act", [(1, 0), (3, 1)])
def test_linear_small_grouped(az, M, in_act):
    """One launch for several independent (x, W, b, y) quadruples of different widths (UNet modulation MLPs,
    ADM FiLM projections): each group against torch.nn.functional.linear."""
    g = torch.Generator().manual_seed(M)
    K, Ns = 64, (12, 256, 40)
    hid = torch.randn(M, len(Ns) * K, generator=g)
    Ws = [torch.randn(n, K, generator=g) / 8 for n in Ns]
    bs = [torch.randn(n, generator=g) for n in Ns]
    dh = dev(hid)
    dW, db = [dev(w) for w in Ws], [dev(b) for b in bs]
    ys = [torch.full((M, n), 7.0, device="cuda") for n in Ns]
    groups = (az.AzLinearGroup * len(Ns))()
    for i, n in enumerate(Ns):
        q = groups[i]
        q.y, q.x, q.W, q.bias = ys[i].data_ptr(), dh.data_ptr() + 4 * i * K, dW[i].data_ptr(), db[i].data_ptr() if i != 1 else None
        q.ldy, q.ldx, q.N, q.K = n, len(Ns) * K, n, K
    gdev = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
    az.call("az_linear_small_grouped_f32", gdev.data_ptr(), len(Ns), max(Ns), M, in_act, 0, az.stream_ptr())
    for i, n in enumerate(Ns):
        x = hid[:, i * K : (i + 1) * K]
        ref = F.linear(F.silu(x) if in_act else x, Ws[i], bs[i] if i != 1 else None)
        assert max_err(ys[i], ref) < 2e-5 * max(1.0, ref.abs().max().item()), i


@pytest.mark.parametrize(
    "B,Cc,H,W,groups",
    [(2, 32, 16, 16, 8), (2, 8, 16, 16, 8), (1, 12, 9, 7, 3), (2, 256, 32, 32, 32), (1, 2048, 8, 8, 32), (3, 64, 64, 64, 32),
     # slices that do not divide 256 threads: 192 float4 chunks (ADM's 768 = 512 + 256), 2 x 192 (1536), 24, 2 x 160 (1280)
     (2, 768, 16, 16, 32), (1, 1536, 8, 8, 32), (2, 96, 16, 16, 8), (1, 1280, 8, 8, 32), (1, 1152, 8, 8, 32)],
)
@pytest.mark.parametrize("affine,mod", [(False, True), (True, True), (True, False)])
def test_groupnorm_mod_silu(az, B, Cc, H, W, groups, affine, mod):
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(B * Cc + H)
    x = torch.randn(B, Cc, H, W, generator=g) * 1.7 + 0.9
    w, b = (torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)) if affine else (None, None)
    a, sh = torch.randn(B, Cc, generator=g), torch.randn(B, Cc, generator=g)
    ref = F.group_norm(x, groups, w, b, eps=1e-5)
    if mod:
        ref = ref * (1 + a[:, :, None, None]) + sh[:, :, None, None]
    ref = F.silu(ref)
    bld = Builder(torch.device("cuda"))
    cs = (Cc + 3) // 4 * 4
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cc, cs, True)
    ab = torch.zeros(B, 2 * cs, device="cuda")
    ab[:, :Cc], ab[:, cs : cs + Cc] = dev(a), dev(sh)
    y = bld.group_norm(
        xa, groups, weight=dev(w) if affine else None, bias=dev(b) if affine else None, scale=ab if mod else None,
        shift=ab if mod else None, shift_off=cs, bstride=2 * cs, act=1,
    )
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, H, W, cs), Cc)
    assert max_err(out, ref) < 2e-5
    assert (y.buf.reshape(B, H, W, cs)[..., Cc:] == 0).all()


@pytest.mark.parametrize("c0,c1,groups", [(512, 256, 32), (1024, 512, 32), (64, 32, 8), (256, 256, 32)])
def test_groupnorm_two_sources(az, c0, c1, groups):
    """GroupNorm over the channel concatenation [x | x1] read in place (plugins/adm/_src/unet.py:631), separate statistics pass."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(c0 + c1)
    B, H, W = 2, 8, 8
    x, x1 = torch.randn(B, c0, H, W, generator=g) + 0.5, torch.randn(B, c1, H, W, generator=g) * 2.0 - 1.0
    w, b = torch.randn(c0 + c1, generator=g), torch.randn(c0 + c1, generator=g)
    ref = F.silu(F.group_norm(torch.cat((x, x1), 1), groups, w, b, eps=1e-5))
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, c0, c0, True)
    xb = Act(to_nhwc(dev(x1)).reshape(-1), B, H, W, c1, c1, True)
    y = bld.group_norm(xa, groups, weight=dev(w), bias=dev(b), act=1, x1=xb)
    assert "az_groupnorm_stats_f32" in [n for _, _, n in bld.tape.ops]
    bld.tape.run()
    assert max_err(from_nhwc(y.buf.reshape(B, H, W, c0 + c1), c0 + c1), ref) < 2e-5


def test_groupnorm_avgpool(az):
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(3)
    B, Cc, H, W = 2, 64, 16, 16
    x = torch.randn(B, Cc, H, W, generator=g)
    w, b = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    ref = F.avg_pool2d(F.silu(F.group_norm(x, 32, w, b, eps=1e-5)), 2, 2)
    bld = Builder(torch.device("cuda"))
    y = bld.group_norm(Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cc, Cc, True), 32, weight=dev(w), bias=dev(b), act=1, pool=1)
    bld.tape.run()
    assert max_err(from_nhwc(y.buf.reshape(B, H // 2, W // 2, Cc), Cc), ref) < 2e-5


@pytest.mark.parametrize("H", [1, 3])
def test_groupnorm_avgpool_along_the_width(az, H):
    """Pooling mode 2: AvgPool1d(2) of a signal held as a one-row image (avg_pool_nd, plugins/adm/_src/nn.py:64-77); rows
    are never mixed."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(4)
    B, Cc, W = 2, 64, 48
    x = torch.randn(B, Cc, H, W, generator=g)
    w, b = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    ref = F.avg_pool2d(F.silu(F.group_norm(x, 32, w, b, eps=1e-5)), (1, 2), (1, 2))
    if H == 1:
        assert torch.equal(ref[:, :, 0], F.avg_pool1d(F.silu(F.group_norm(x[:, :, 0], 32, w, b, eps=1e-5)), 2, 2))
    bld = Builder(torch.device("cuda"))
    y = bld.group_norm(Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cc, Cc, True), 32, weight=dev(w), bias=dev(b), act=1, pool=2)
    assert (y.H, y.W) == (H, W // 2)
    bld.tape.run()
    assert max_err(from_nhwc(y.buf.reshape(B, H, W // 2, Cc), Cc), ref) < 2e-5


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("Cc", [5, 64, 768, 1000])
def test_rownorm_mod(az, kind, Cc):
    from azula_amd.engine import Act, Builder
    from oracle import nets

    g = torch.Generator().manual_seed(Cc)
    B, H, W = 2, 3, 5
    x = torch.randn(B, Cc, H, W, generator=g) * 2 + 0.5
    a, sh = torch.randn(B, Cc, generator=g), torch.randn(B, Cc, generator=g)
    n = nets.layer_norm_unbiased(x, dim=1) if kind == 0 else nets.rms_norm(x, dim=1)
    ref = n * (1 + a[:, :, None, None]) + sh[:, :, None, None]
    cs = (Cc + 3) // 4 * 4
    ab = torch.zeros(B, 2 * cs, device="cuda")
    ab[:, :Cc], ab[:, cs : cs + Cc] = dev(a), dev(sh)
    bld = Builder(torch.device("cuda"))
    y = bld.row_norm(Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cc, cs, True), kind, scale=ab, shift=ab, shift_off=cs, bstride=2 * cs)
    bld.tape.run()
    assert max_err(from_nhwc(y.buf.reshape(B, H, W, cs), Cc), ref) < 2e-5


CONV_CASES = [
    # B, Cin, Cout, H, W, ks, stride
    (2, 3, 16, 16, 16, 3, 1),
    (1, 32, 32, 8, 8, 3, 1),
    (2, 40, 24, 9, 7, 3, 1),
    (2, 16, 32, 16, 16, 3, 2),
    (1, 5, 7, 15, 15, 3, 2),
    (2, 64, 48, 8, 8, 1, 1),
    (1, 256, 256, 16, 16, 3, 1),
    (1, 130, 260, 12, 12, 3, 1),
    (4, 8, 3, 32, 32, 3, 1),
]


def conv_tol(cin, ks, wino=False):
    """Direct: exact fp32 fmaf chains.  F(2x2,3x3) only adds/subtracts (3x).  F(4x4,3x3) multiplies by up to 8
    and its filter transform by 1/24: ~20x the rounding error of F(2x2) (stated in include/azula_amd.h)."""
    return (3e-6 * math.sqrt(cin * ks * ks) + 1e-5) * {False: 1, True: 3, 4: 40, "x3": 1, "wx3": 3, "h2": 1, "wh2": 3}[wino]


WINO_NAME = {False: "az_conv2d_f32", True: "az_conv2d_winograd_f32", 4: "az_conv2d_winograd4_f32", "x3": "az_conv2d_x3_f32",
             "wx3": "az_conv2d_winograd_x3_f32",  # "wx3": Winograd with its frequency GEMMs on the bf16 pipe (csrc/wino_x3.hip)
             "h2": "az_conv2d_f16x2_f32", "wh2": "az_conv2d_winograd_f16x2_f32"}  # the f16x2 forms of the two (2 half pieces, 3 products)


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,stride", CONV_CASES)
@pytest.mark.parametrize("splitk", [0, 3])
@pytest.mark.parametrize("wino", [False, True, 4, "x3", "wx3", "h2", "wh2"])
def test_conv2d_basic(az, B, Cin, Cout, H, W, ks, stride, splitk, wino):
    if wino in (True, 4, "wx3", "wh2") and (ks != 3 or stride != 1):
        pytest.skip("Winograd is the stride-1 3x3 path")
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(Cin * Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=ks // 2)
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, (Cin + 3) // 4 * 4, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, stride=stride, winograd=wino)
    assert bld.tape.ops[-1][2] == WINO_NAME[wino]
    if splitk:
        a = bld.tape.keep[-1]
        a.splitk = splitk
        bld._ws_need = max(bld._ws_need, splitk * B * y.H * y.W * y.cs)
        bld._ws_users.append(a)
    bld.finish()
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, y.H, y.W, y.cs), Cout)
    assert out.shape == ref.shape
    assert max_err(out, ref) < conv_tol(Cin, ks, wino), max_err(out, ref)
    assert (y.buf.reshape(B, y.H, y.W, y.cs)[..., Cout:] == 0).all()


@pytest.mark.parametrize("wino", [False, True, "x3", "wx3", "h2", "wh2"])
def test_conv2d_output_beyond_the_infinity_cache(az, wino):
    """An output of 256 MiB (1 x 512 x 512 x 256 fp32): from this size on the plain epilogue stores with the non-temporal hint
    (conv_shared.h: stream_out -- the output cannot stay in the Infinity Cache for its consumer).  Same bound as every other
    3 x 3 case; the store policy must not change a value."""
    from azula_amd.engine import Act, Builder

    B, Cin, Cout, H, W = 1, 32, 256, 512, 512
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, winograd=wino)
    assert bld.tape.ops[-1][2] == WINO_NAME[wino]
    assert B * y.H * y.W * y.cs * 4 >= 256 << 20
    bld.finish()
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, y.H, y.W, y.cs), Cout)
    assert max_err(out, ref) < conv_tol(Cin, 3, wino), max_err(out, ref)


@pytest.mark.parametrize("wino", [False, True, 4, "x3", "wx3", "h2", "wh2"])
def test_conv2d_random_shapes(az, wino):
    """Seeded sweep over ragged shapes (odd sizes, channel counts off the 4 / 8 / 32 grids, batch 1-3, optional second
    source with nearest-x2 upsampling, SiLU / gate / residual) for the three 3x3 stride-1 algorithms."""
    import random

    from azula_amd.engine import Act, Builder

    rnd = random.Random(1234 + {False: 0, True: 7, 4: 4, "x3": 3, "wx3": 7, "h2": 3, "wh2": 7}[wino])
    g = torch.Generator().manual_seed(99)
    for case in range(10):
        B = rnd.randint(1, 3)
        H, W = rnd.randint(3, 21), rnd.randint(3, 21)
        C0, Cout = rnd.choice([3, 5, 8, 12, 20, 33, 40]), rnd.choice([3, 6, 8, 17, 32, 70])
        two = rnd.random() < 0.4
        C1 = rnd.choice([4, 7, 16]) if two else 0
        act = rnd.choice([0, 1])
        x0 = torch.randn(B, C0, H, W, generator=g)
        w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1))
        b = torch.randn(Cout, generator=g)
        gate, res = torch.randn(B, Cout, generator=g), torch.randn(B, Cout, H, W, generator=g)
        src = x0
        bld = Builder(torch.device("cuda"))
        a0 = Act(to_nhwc(dev(x0)).reshape(-1), B, H, W, C0, (C0 + 3) // 4 * 4, True)
        kw = {}
        if two:
            h1, w1 = (H + 1) // 2, (W + 1) // 2
            x1 = torch.randn(B, C1, h1, w1, generator=g)
            up = F.interpolate(x1, scale_factor=(2.0, 2.0), mode="nearest")[:, :, :H, :W]
            src = torch.cat((x0, up), 1)
            a1 = Act(to_nhwc(dev(x1)).reshape(-1), B, h1, w1, C1, (C1 + 3) // 4 * 4, True)
            kw = dict(src1=a1, up1=1, hin=H, win=W)
        ref = F.conv2d(src, w, b, padding=1)
        ref = res + gate[:, :, None, None] * (F.silu(ref) if act else ref)
        cs_o = (Cout + 3) // 4 * 4
        gpad = torch.zeros(B, cs_o)
        gpad[:, :Cout] = gate
        ra = Act(to_nhwc(dev(res)).reshape(-1), B, H, W, Cout, cs_o, True)
        y = bld.conv(a0, bld.pack_conv(dev(w), dev(b), cin0=C0), Cout, act=act, gate=dev(gpad), gate_bstride=cs_o, res=ra,
                     winograd=wino, **kw)
        bld.finish()
        bld.tape.run()
        out = from_nhwc(y.buf.reshape(B, H, W, cs_o), Cout)
        err = max_err(out, ref)
        assert err < conv_tol(C0 + C1, 3, wino) * max(1.0, ref.abs().max().item()), (case, B, H, W, C0, C1, Cout, act, err)


@pytest.mark.parametrize("wino", [False, True, "wx3", "wh2"])
@pytest.mark.parametrize("shift", [-2, -1, 1, 2])
@pytest.mark.parametrize("Cin,H,W", [(32, 16, 16), (64, 12, 20), (20, 9, 7)])
def test_conv2d_depth_tap_between_poisoned_neighbours(az, wino, shift, Cin, H, W):
    """One depth tap of a 3-D convolution (AzConvArgs.depth / depth_shift: image b reads plane b + shift of its own volume, a
    plane outside the volume reads zeros) with NaN-filled memory directly in front of and behind the source: the buffer
    descriptors must stay inside the allocation whatever the shift (their base used to move by `shift` planes, so
    only the per-lane masks stood between a tap and its neighbours).  32- / 64-channel cases take the hand-scheduled streams."""
    from azula_amd.engine import Act, Builder

    D, V, Cout = 3, 2, 64
    B = D * V
    g = torch.Generator().manual_seed(Cin + H + shift)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    full = F.conv2d(x, w, None, padding=1)
    ref = b[None, :, None, None].expand(B, Cout, H, W).clone()
    for i in range(B):
        if 0 <= i % D + shift < D:
            ref[i] += full[i + shift]
    cs = (Cin + 3) // 4 * 4
    n = B * H * W * cs
    guard = 2 * H * W * cs + 4096
    arena = torch.full((n + 2 * guard,), float("nan"), device="cuda")
    arena[guard:guard + n] = to_nhwc(dev(x)).reshape(-1)
    bld = Builder(torch.device("cuda"))
    xa = Act(arena[guard:guard + n], B, H, W, Cin, cs, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, winograd=wino, depth=(D, shift))
    bld.finish()
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, H, W, y.cs), Cout)
    assert torch.isfinite(out).all()
    assert max_err(out, ref) < conv_tol(Cin, 3, wino), max_err(out, ref)


@pytest.mark.parametrize("wino", [False, True, "wx3", "wh2"])
@pytest.mark.parametrize("Cin,Cout,ks", [(32, 64, 1), (20, 24, 3), (64, 256, 1)])
def test_conv2d_swiglu_epilogue(az, wino, Cin, Cout, ks):
    """AzConvArgs.act = 4: y[c] = x[2c] * silu(x[2c+1]) applied to the convolution's output in its epilogue (half the
    channels come out) -- azula/nn/layers.py:107-110 behind a Linear, JiT's SwiGLUFFN."""
    if wino and ks != 3:
        pytest.skip("Winograd is the stride-1 3x3 path")
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(Cin + Cout)
    B, H, W = 2, 9, 7
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    full = F.conv2d(x, w, b, padding=ks // 2)
    ref = full[:, 0::2] * F.silu(full[:, 1::2])
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, (Cin + 3) // 4 * 4, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, act=4, winograd=wino)
    bld.finish()
    bld.tape.run()
    assert (y.C, y.cs) == (Cout // 2, Cout // 2)
    out = from_nhwc(y.buf.reshape(B, H, W, y.cs), Cout // 2)
    assert max_err(out, ref) < 2 * conv_tol(Cin, ks, wino) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("wino", [False, True, "x3", "wx3", "h2", "wh2"])
@pytest.mark.parametrize("Cin,Cout,ks,splitk", [(32, 64, 1, 0), (20, 24, 3, 0), (64, 128, 3, 2)])
def test_conv2d_silu_of_the_sum_with_the_residual(az, wino, Cin, Cout, ks, splitk):
    """AzConvArgs.act = 6: y = silu(conv + bias + res), in place on the residual operand -- the last depth tap of a Conv3d -> SiLU
    pair (nn/unet3d.py: no activation pass of its own)."""
    if wino in (True, "wx3", "wh2") and ks != 3:
        pytest.skip("Winograd is the stride-1 3x3 path")
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(Cin + Cout + ks)
    B, H, W = 2, 10, 6
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, H, W, generator=g)
    ref = F.silu(F.conv2d(x, w, b, padding=ks // 2) + r)
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, (Cin + 3) // 4 * 4, True)
    acc = Act(to_nhwc(dev(r)).reshape(-1).clone(), B, H, W, Cout, Cout, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, act=6, res=acc, out=acc, winograd=wino)
    if splitk:
        d = [k for k in bld.tape.keep if hasattr(k, "_flops")][-1]
        d.splitk = splitk
        bld._ws_need = max(bld._ws_need, splitk * B * H * W * Cout)
        if d not in bld._ws_users:
            bld._ws_users.append(d)
    bld.finish()
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, H, W, y.cs), Cout)
    assert max_err(out, ref) < 2 * conv_tol(Cin, ks, bool(wino) and wino not in ("x3", "h2")) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode", [True, "wx3", "wh2"])
def test_winograd_stream_fuzz(az, mode):
    """The hand-scheduled K loop (wino_kloop.inc) over 40 seeded cases that move every event of the stream around: 1 .. 24
    eight-channel stages (first / steady / second-to-last / last iteration bodies), two sources whose switch falls on any
    stage, half chunks at the end of either source, split-K slices that start in either source, nearest-x2 upsampling of the
    second source, zero and circular padding, ragged tile blocks."""
    import random

    from azula_amd.engine import Act, Builder

    rnd = random.Random(2025)
    g = torch.Generator().manual_seed(2025)
    for case in range(40):
        B = rnd.randint(1, 2)
        H, W = rnd.randint(4, 24), rnd.randint(4, 24)
        C0 = rnd.choice([4, 8, 12, 20, 36, 60, 64, 100, 132])
        two = rnd.random() < 0.5
        C1 = rnd.choice([4, 8, 12, 28, 64, 68]) if two else 0
        Cout = rnd.choice([8, 24, 64, 72, 130])
        periodic = rnd.random() < 0.3
        up = two and rnd.random() < 0.5 and not periodic
        splitk = rnd.choice([0, 0, 2, 3, 5])
        x0 = torch.randn(B, C0, H, W, generator=g)
        w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1))
        b = torch.randn(Cout, generator=g)
        src = x0
        bld = Builder(torch.device("cuda"))
        a0 = Act(to_nhwc(dev(x0)).reshape(-1), B, H, W, C0, (C0 + 3) // 4 * 4, True)
        kw = {}
        if two:
            h1, w1 = ((H + 1) // 2, (W + 1) // 2) if up else (H, W)
            x1 = torch.randn(B, C1, h1, w1, generator=g)
            full = F.interpolate(x1, scale_factor=(2.0, 2.0), mode="nearest")[:, :, :H, :W] if up else x1
            src = torch.cat((x0, full), 1)
            a1 = Act(to_nhwc(dev(x1)).reshape(-1), B, h1, w1, C1, (C1 + 3) // 4 * 4, True)
            kw = dict(src1=a1, up1=1 if up else 0, hin=H, win=W)
        ref = F.conv2d(F.pad(src, (1, 1, 1, 1), mode="circular"), w, b) if periodic else F.conv2d(src, w, b, padding=1)
        y = bld.conv(a0, bld.pack_conv(dev(w), dev(b), cin0=C0), Cout, winograd=mode, periodic=periodic, **kw)
        assert bld.tape.ops[-1][2] == WINO_NAME[mode]
        if splitk:
            a = bld.tape.keep[-1]
            a.splitk = splitk
            bld._ws_need = max(bld._ws_need, splitk * B * H * W * y.cs)
            bld._ws_users.append(a)
        bld.finish()
        bld.tape.run()
        out = from_nhwc(y.buf.reshape(B, H, W, y.cs), Cout)
        err = max_err(out, ref)
        assert err < conv_tol(C0 + C1, 3, True) * max(1.0, ref.abs().max().item()), (case, B, H, W, C0, C1, Cout, periodic, up, splitk, err)


def test_direct_taps_stream_fuzz(az):
    """The taps variant of the direct kernel's hand-scheduled K loop (igemm_kloop.inc, IGEMM_KLOOP_TAPS_ASM: k x k convolutions
    at any stride, one source, zero padding, channels in whole 32-channel K tiles, >= 16 stages per split-K slice) over 30
    seeded cases: 3 x 3 and 5 x 5, strides 1 .. 3, ragged pixel / channel tiles, images smaller than the window, split-K
    slices that start in the middle of a tap."""
    import random

    from azula_amd.engine import Act, Builder

    rnd = random.Random(4242)
    g = torch.Generator().manual_seed(4242)
    for case in range(30):
        ks = rnd.choice([3, 3, 5])
        Cin = 32 * rnd.choice([2, 3, 5, 8] if ks == 3 else [1, 2, 3])
        stride = rnd.choice([1, 2, 2, 3])
        B = rnd.randint(1, 3)
        H, W = rnd.randint(2, 40), rnd.randint(2, 40)
        Cout = rnd.choice([8, 24, 128, 136, 260])
        nk = ks * ks * (Cin // 32)
        splitk = rnd.choice([s_ for s_ in (1, 2, 3) if nk // s_ >= 16])
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(ks * ks * Cin)
        b = torch.randn(Cout, generator=g)
        ref = F.conv2d(x, w, b, padding=ks // 2, stride=stride)
        bld = Builder(torch.device("cuda"))
        xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, stride=stride, winograd=False)
        a = bld.tape.keep[-1]
        assert bld.tape.ops[-1][2] == "az_conv2d_f32"
        a.splitk = splitk
        if splitk > 1:
            bld._ws_need = max(bld._ws_need, splitk * B * y.H * y.W * y.cs)
            if a not in bld._ws_users:
                bld._ws_users.append(a)
        bld.finish()
        bld.tape.run()
        out = from_nhwc(y.buf.reshape(B, y.H, y.W, y.cs), Cout)
        err = max_err(out, ref)
        assert err < conv_tol(Cin, ks, False) * max(1.0, ref.abs().max().item()), (case, B, H, W, Cin, Cout, ks, stride, splitk, err)


@pytest.mark.parametrize("cin", [1, 2, 3, 4])
@pytest.mark.parametrize("B,H,W,cout,periodic", [(2, 16, 32, 64, False), (3, 9, 37, 24, False), (1, 40, 70, 320, True), (2, 5, 3, 8, True)])
def test_conv2d_stem(az, cin, B, H, W, cout, periodic):
    """az_conv2d_stem_f32: the first 3 x 3 convolution reading its <= 4 input channels planar (the latent's own layout), NHWC
    out, with the GroupNorm moments of its output -- against torch.conv2d and torch.group_norm (ragged tiles, more than 256
    output channels, circular padding)."""
    from azula_amd.engine import Builder

    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g) + 3.0
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="circular"), w, b) if periodic else F.conv2d(x, w, b, padding=1)
    groups = 2  # (whole channel quads per group: the moments come from the stem kernel)
    gw, gb = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    ref_n = F.group_norm(ref, groups, gw, gb, eps=1e-5)
    bld = Builder(torch.device("cuda"))
    y = bld.conv_stem(dev(x).contiguous(), B, cin, H, W, bld.pack_conv(dev(w), dev(b)), cout, periodic=periodic, gn_stats=True)
    n = bld.group_norm(y, groups, weight=dev(gw), bias=dev(gb))
    bld.finish()
    names = [nm for _, _, nm in bld.tape.ops]
    assert names[0] == "az_conv2d_stem_f32" and "az_groupnorm_stats_f32" not in names
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, H, W, cout), cout)
    assert max_err(out, ref) < 2e-6 * max(1.0, ref.abs().max().item())
    assert max_err(from_nhwc(n.buf.reshape(B, H, W, cout), cout), ref_n) < 2e-5


@pytest.mark.parametrize("asm", ["1", "0", "wx3"])
@pytest.mark.parametrize("in_act", [0, 1])
def test_winograd_input_affine(az, asm, in_act, monkeypatch):
    """AzConvArgs.in_affine: conv(act(x * scale[b, c] + shift[b, c])) with the affine evaluated inside the Winograd gather
    (the GroupNorm apply pass, fused: azula/nn/unet.py:85-92).  Zero padding pads the NORMALISED tensor; circular padding
    wraps it.  24 seeded cases: 1 .. 20 stages, ragged tile blocks, tile blocks that span several images, split-K slices, both
    K loops (the hand-scheduled stream / the C++ loop)."""
    import random

    from azula_amd.engine import Act, Builder

    mode = "wx3" if asm == "wx3" else True  # ("wx3": the bf16-pipe kernel, csrc/wino_x3.hip)
    if mode is True:
        monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        monkeypatch.setenv("AZ_WINOGRAD_ASM", asm)
    rnd = random.Random(77 + in_act)
    g = torch.Generator().manual_seed(77 + in_act)
    for case in range(24):
        B = rnd.randint(1, 5)
        H, W = rnd.choice([(4, 4), (6, 10), (16, 16), (9, 7), (24, 20), (32, 32)])
        C0 = 8 * rnd.choice([1, 2, 3, 5, 8, 20])
        Cout = rnd.choice([8, 24, 64, 72, 130])
        periodic = rnd.random() < 0.3
        splitk = rnd.choice([0, 0, 2, 3])
        x = torch.randn(B, C0, H, W, generator=g) * 2.0 + 1.0
        sc, sh = torch.randn(B, C0, generator=g), torch.randn(B, C0, generator=g) * 3.0  # (shift != 0: padding must stay zero)
        w = torch.randn(Cout, C0, 3, 3, generator=g) / math.sqrt(9 * C0)
        b = torch.randn(Cout, generator=g)
        n = x * sc[:, :, None, None] + sh[:, :, None, None]
        if in_act:
            n = F.silu(n)
        ref = F.conv2d(F.pad(n, (1, 1, 1, 1), mode="circular"), w, b) if periodic else F.conv2d(n, w, b, padding=1)
        bld = Builder(torch.device("cuda"))
        a0 = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, C0, C0, True)
        a0.affine = (dev(torch.cat((sc.reshape(-1), sh.reshape(-1)))), in_act)
        y = bld.conv(a0, bld.pack_conv(dev(w), dev(b)), Cout, winograd=mode, periodic=periodic)
        assert [nm for _, _, nm in bld.tape.ops] == [WINO_NAME[mode]]  # no apply pass
        if splitk:
            a = bld.tape.keep[-1]
            a.splitk = splitk
            bld._ws_need = max(bld._ws_need, splitk * B * H * W * y.cs)
            bld._ws_users.append(a)
        bld.finish()
        bld.tape.run()
        out = from_nhwc(y.buf.reshape(B, H, W, y.cs), Cout)
        err = max_err(out, ref)
        assert err < conv_tol(C0, 3, True) * max(1.0, ref.abs().max().item(), n.abs().max().item()), (case, B, H, W, C0, Cout, periodic, splitk, err)
    # every other entry point rejects the field instead of ignoring it
    a = bld.tape.keep[-1]
    a.splitk = 1
    assert az.lib().az_conv2d_f32(C.byref(a), az.stream_ptr()) == -4  # AZ_E_UNSUPPORTED


@pytest.mark.parametrize("wino", [False, True, 4, "x3", "wx3", "h2", "wh2"])
def test_conv2d_concat_upsample_narrow_gate_res(az, wino):
    """cat((y, upsample(x)[narrowed])) -> conv -> x0 + c * silu-free epilogue, as azula/nn/unet.py:253-257,93."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(11)
    B, Cy, Cx, Cout, H, W = 2, 12, 20, 12, 15, 13
    y = torch.randn(B, Cy, H, W, generator=g)
    x = torch.randn(B, Cx, 8, 7, generator=g)
    w = torch.randn(Cout, Cy + Cx, 3, 3, generator=g) / math.sqrt(9 * (Cy + Cx))
    b = torch.randn(Cout, generator=g)
    gate = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    up = F.interpolate(x, scale_factor=(2.0, 2.0), mode="nearest")[:, :, :H, :W]
    ref = res + gate[:, :, None, None] * F.silu(F.conv2d(torch.cat((y, up), 1), w, b, padding=1))
    bld = Builder(torch.device("cuda"))
    ya = Act(to_nhwc(dev(y)).reshape(-1), B, H, W, Cy, 12, True)
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, 8, 7, Cx, 20, True)
    ra = Act(to_nhwc(dev(res)).reshape(-1), B, H, W, Cout, 12, True)
    out = bld.conv(ya, bld.pack_conv(dev(w), dev(b), cin0=Cy), Cout, src1=xa, up1=1, hin=H, win=W, act=1,
                   gate=dev(gate), gate_bstride=Cout, res=ra, winograd=wino)
    bld.finish()
    bld.tape.run()
    assert max_err(from_nhwc(out.buf.reshape(B, H, W, 12), Cout), ref) < conv_tol(Cy + Cx, 3, wino)


@pytest.mark.parametrize("wino", [False, True, 4, "x3", "wx3", "h2", "wh2"])
def test_conv2d_nchw_output_and_res_up(az, wino):
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(12)
    B, Cin, Cout, H, W = 2, 16, 3, 12, 12
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 12
    b = torch.randn(Cout, generator=g)
    bld = Builder(torch.device("cuda"))
    dst = torch.empty(B, Cout, H, W, device="cuda")
    x0a = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, 16, True)  # keep alive until the tape has run
    bld.conv(x0a, bld.pack_conv(dev(w), dev(b)), Cout, dst_nchw=dst, winograd=wino)
    # ADM up-block shape: conv over upsampled input + upsampled identity residual
    w2 = torch.randn(Cin, Cin, 3, 3, generator=g) / 12
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, 16, True)
    up = bld.conv(xa, bld.pack_conv(dev(w2), None), Cin, up0=1, res=xa, res_up=1, winograd=wino)
    bld.finish()
    bld.tape.run()
    tol = conv_tol(Cin, 3, wino)
    assert max_err(dst, F.conv2d(x, w, b, padding=1)) < tol
    xu = F.interpolate(x, scale_factor=2, mode="nearest")
    assert max_err(from_nhwc(up.buf.reshape(B, 2 * H, 2 * W, 16), Cin), xu + F.conv2d(xu, w2, None, padding=1)) < tol


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(1, 16, 3, 16, 16), (2, 48, 1, 37, 21), (3, 256, 3, 19, 50), (1, 32, 4, 5, 3), (2, 64, 2, 33, 33)])
@pytest.mark.parametrize("nchw", [False, True])
def test_conv2d_narrow_output_kernel(az, B, Cin, Cout, H, W, nchw):
    """cout_s == 4, 3x3 stride 1, one source with Cin % 16 == 0: az_conv2d_f32 runs conv_head_kernel (the image head
    of the UNets) -- ragged maps (partial 16 x 16 tiles), 1..4 real channels, both destinations, full epilogue."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
    if nchw:
        dst = torch.full((B, Cout, H, W), float("nan"), device="cuda")
        bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, dst_nchw=dst, winograd=False)
        ref = F.conv2d(x, w, b, padding=1)
    else:
        gate, res = torch.randn(B, Cout, generator=g), torch.randn(B, Cout, H, W, generator=g)
        gpad = torch.zeros(B, 4)
        gpad[:, :Cout] = gate
        ra = Act(to_nhwc(dev(res), 4).reshape(-1), B, H, W, Cout, 4, True)
        y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, act=1, gate=dev(gpad), gate_bstride=4, res=ra, winograd=False)
        ref = res + gate[:, :, None, None] * F.silu(F.conv2d(x, w, b, padding=1))
    assert bld.tape.ops[-1][2] == "az_conv2d_f32"
    bld.finish()
    bld.tape.run()
    out = dst if nchw else from_nhwc(y.buf.reshape(B, H, W, 4), Cout)
    assert max_err(out, ref) < conv_tol(Cin, 3) * max(1.0, ref.abs().max().item())


def test_conv2d_x3_accuracy(az, monkeypatch):
    """fp32 operands as 3 x bf16 pieces / 6 partial products (az_conv2d_x3_f32) against an fp64 reference, next to the
    fp32-MFMA direct kernel and the two Winograd forms (fp32 stream, frequency GEMMs on the bf16 pipe) on the same layer (Cin = 256,
    K = 2304): the split path must be at the accuracy level of the direct fp32 kernel (<= 2x its error) and no worse than the Winograd
    kernel; the x3 Winograd kernel at the level of the fp32 Winograd stream.  (The library's own
    tile plan: forcing this 8-tile grid onto unsplit 256 x 256 tiles makes one fp32 chain of all 2304 products per output.)"""
    from azula_amd.engine import Act, Builder

    monkeypatch.delenv("AZ_X3_BIG", raising=False)

    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W = 1, 256, 128, 32, 32
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))  # mixed channel scales
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    errs = {}
    for mode in (False, True, "x3", "wx3", "h2", "wh2"):
        bld = Builder(torch.device("cuda"))
        xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, winograd=mode)
        assert bld.tape.ops[-1][2] == WINO_NAME[mode]
        bld.finish()
        bld.tape.run()
        out = from_nhwc(y.buf.reshape(B, H, W, Cout), Cout).double().cpu()
        e = (out - ref).abs()
        errs[mode] = (e.max().item(), e.pow(2).mean().sqrt().item())
    print("conv error vs fp64 (max, rms):", errs)
    assert errs["x3"][1] <= 2.0 * errs[False][1] and errs["x3"][0] <= 2.0 * errs[False][0], errs
    assert errs["x3"][1] <= errs[True][1], errs
    # the Winograd form with its frequency GEMMs as exact 3 x bf16 splits: the same transforms as the fp32 Winograd stream, so its
    # error is that stream's (the transforms' fp32 adds dominate both), not worse by more than the rounding of the accumulation order
    assert errs["wx3"][1] <= 1.25 * errs[True][1] and errs["wx3"][0] <= 1.5 * errs[True][0], errs
    # f16x2 (two half pieces per operand, three products, one fp32 accumulator): each product carries ~2^-22 of relative error, far
    # below what the fp32 accumulation of 2304 terms adds to every form -- at the level of the fp32 MFMA kernel / the fp32 Winograd stream
    assert errs["h2"][1] <= 1.25 * errs[False][1] and errs["h2"][0] <= 1.5 * errs[False][0], errs
    assert errs["wh2"][1] <= 1.25 * errs[True][1] and errs["wh2"][0] <= 1.5 * errs[True][0], errs


@pytest.mark.parametrize("mode", ["h2d", "wh2d"])
@pytest.mark.parametrize("two", [False, True])
def test_f16x2_dynamic_scale_has_no_range(az, mode, two):
    """AzConvArgs.in_absmax0 / in_absmax1 (az_absmax_f32 over the sources): the f16x2 kernels scale their activation operand by the
    power of two that fits the sources' largest magnitude -- any finite fp32 input goes through at fp32-level relative error, from
    1e-20 to 1e20 (the fixed scale's range ends at ~1e6: test_f16x2_domain), also for a channel concatenation of two sources of
    very different size; a non-finite element poisons only the outputs it reaches."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(8)
    B, C0, C1, Cout, H, W = 2, 64, 32 if two else 0, 96, 20, 24
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1))

    def run(x0, x1):
        bld = Builder(torch.device("cuda"))
        a0 = Act(to_nhwc(dev(x0)).reshape(-1), B, H, W, C0, C0, True)
        a1 = Act(to_nhwc(dev(x1)).reshape(-1), B, H, W, C1, C1, True) if two else None
        y = bld.conv(a0, bld.pack_conv(dev(w), None, cin0=C0 if two else None), Cout, src1=a1, winograd=mode)
        names = [n for _, _, n in bld.tape.ops]
        assert names.count("az_absmax_f32") == (2 if two else 1) and names[-1] == WINO_NAME[mode[:-1]], names
        bld.finish()
        bld.tape.run()
        return from_nhwc(y.buf.reshape(B, H, W, Cout), Cout).cpu()

    for s0, s1 in ((1.0, 1.0), (1e7, 3.0), (1e-20, 1e-20), (1e20, 1e15), (3e-5, 40.0)):
        x0 = torch.randn(B, C0, H, W, generator=g) * s0
        x1 = torch.randn(B, max(C1, 1), H, W, generator=g) * s1
        ref = F.conv2d((torch.cat([x0, x1], 1) if two else x0).double(), w.double(), None, padding=1)
        out = run(x0, x1)
        err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
        print(mode, two, s0, s1, "relative error", err)
        assert torch.isfinite(out).all() and err < 3e-6, (s0, s1, err)
    x0 = torch.randn(B, C0, H, W, generator=g)
    x0[1, 5, 9, 9] = float("inf")
    out = run(x0, torch.randn(B, max(C1, 1), H, W, generator=g))
    assert not torch.isfinite(out[1, :, 8:11, 8:11]).any() and torch.isfinite(out[0]).all()


@pytest.mark.parametrize("n", [1, 3, 4, 1000, 65536 * 3 + 2, 4 * 1024 * 1024])
def test_absmax(az, n):
    """az_absmax_f32: AZ_ABSMAX_SLOTS partial maxima whose maximum is max |x| (every slot written; NaN elements ignored)."""
    from azula_amd import _lib

    g = torch.Generator().manual_seed(n)
    x = torch.randn(n + 4, generator=g)
    x[n // 2] = -123.5
    x[n:] = 1e9  # (past the end: must not be read into the result)
    xd = dev(x)
    slots = torch.full((256,), float("nan"), device="cuda")
    _lib.call("az_absmax_f32", slots.data_ptr(), xd.data_ptr(), n, _lib.stream_ptr())
    assert torch.isfinite(slots).all() and slots.max().item() == x[:n].abs().max().item()


@pytest.mark.parametrize("mode", [True, "wh2", "x3"])
def test_absmax_from_the_producers_moments(az, mode, monkeypatch):
    """az_absmax_from_moments_f32: the GroupNorm partial moments a convolution leaves of its output (AzConvArgs.gn_quads: Winograd
    epilogue or split-K combine) bound its largest magnitude from above -- |x| <= |mean| + sqrt(M2) per record -- within sqrt(n) of
    the true maximum, so that a consumer's f16x2 activation scale needs no pass over the tensor (engine.Builder.absmax_of)."""
    from azula_amd import _lib, engine
    from azula_amd.engine import Act, Builder

    monkeypatch.setattr(engine, "F16X2_MOMENTS", True)  # (whatever AZ_F16X2_MOMENTS says: this test is about that path)
    g = torch.Generator().manual_seed(11)
    B, Cin, Cout, H, W = 2, 64, 128, 32, 32
    x = torch.randn(B, Cin, H, W, generator=g) * 3
    x[1, 7, 5, 9] = 250.0  # (an outlier: the bound must cover it)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), None), Cout, winograd=mode, gn_stats=True)
    if mode == "x3":  # the direct kernels leave moments from their split-K combine only
        d = [k for k in bld.tape.keep if hasattr(k, "_flops")][-1]
        assert d.splitk > 1 or y.gn_quads is None
    if y.gn_quads is None:
        pytest.skip("this launch form leaves no moments for this shape")
    slots = bld.absmax_of(y)
    assert bld.tape.ops[-1][2] == "az_absmax_from_moments_f32"
    bld.finish()
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, H, W, Cout), Cout)
    true_max, bound = out.abs().max().item(), slots.max().item()
    n = out.numel() * 4 // y.gn_quads[0].numel()  # elements per record
    print(mode, "max |y|", true_max, "bound from the moments", bound, "elements per record", n)
    assert true_max <= bound <= true_max * math.sqrt(n) * 1.01 + 1e-6
    # and the measured form on the same tensor
    exact = torch.empty(256, device="cuda")
    _lib.call("az_absmax_f32", exact.data_ptr(), y.ptr, y.buf.numel(), _lib.stream_ptr())
    assert exact.max().item() == true_max


@pytest.mark.parametrize("mode", ["h2", "wh2"])
def test_f16x2_domain(az, mode):
    """The STATED domain of the f16x2 split (include/azula_amd.h, csrc/common.h: az_split2h): activations of any magnitude below
    65520 / AZ_F16X2_IN_SCALE ~ 1.0e6 (4-pixel sums of them in the Winograd form).  Inside it the relative error against fp64 is
    at fp32 level for tensors of scale 1e-3 ... 1e4 (the weights are rescaled at pack time whatever their size: 1e-6 ... 1e3 here);
    tensors of scale 1e-6 keep an ABSOLUTE operand error of 2^-36 * 16 = 2.3e-10 (the half pieces are subnormal, which the matrix
    pipe honours: no flush to zero); beyond the domain the outputs that depend on the offending activation are NaN, never finite garbage."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(6)
    B, Cin, Cout, H, W = 1, 64, 64, 16, 16

    def run(x, w):
        bld = Builder(torch.device("cuda"))
        xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        y = bld.conv(xa, bld.pack_conv(dev(w), None), Cout, winograd=mode)
        assert bld.tape.ops[-1][2] == WINO_NAME[mode]
        bld.finish()
        bld.tape.run()
        return from_nhwc(y.buf.reshape(B, H, W, Cout), Cout).cpu()

    w0 = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    for xs, ws, rel in ((1.0, 1.0, 3e-6), (1e-3, 1e-6, 3e-6), (1e4, 1e3, 3e-6), (30.0, 1.0, 3e-6), (1e-6, 1.0, 3e-4)):
        x = torch.randn(B, Cin, H, W, generator=g) * xs
        x[:, ::5] *= 1e-3  # (channels three decades below the rest: their pieces sit further down the half range)
        w = w0 * ws
        ref = F.conv2d(x.double(), w.double(), None, padding=1)
        err = (run(x, w).double() - ref).abs().max().item()
        print(mode, xs, ws, "relative error", err / ref.abs().max().item())
        assert err < rel * ref.abs().max().item(), (xs, ws, err, ref.abs().max().item())
    # exact zeros stay exact zeros (a zero-initialised layer, zero padding)
    assert run(torch.randn(B, Cin, H, W, generator=g), torch.zeros_like(w0)).abs().max().item() == 0.0
    # outside the domain: one activation of 4e6 -> every output its 3x3 window reaches is NaN; the rest of the map is untouched
    x = torch.randn(B, Cin, H, W, generator=g)
    x[0, 3, 8, 8] = 4.0e6
    out = run(x, w0)
    ref = F.conv2d(x, w0, None, padding=1)
    assert not torch.isfinite(out[0, :, 7:10, 7:10]).any()
    far = torch.ones(B, Cout, H, W, dtype=torch.bool)
    far[:, :, 5:12, 5:12] = False  # (the Winograd tiles touching that pixel turn NaN as a whole)
    assert torch.isfinite(out[far]).all() and max_err(out[far], ref[far]) < conv_tol(Cin, 3, True)


@pytest.mark.parametrize("mode", ["x3", "wx3"])
def test_x3_split_domain(az, mode):
    """The stated domain of the exact 3 x bf16 split (csrc/common.h, include/azula_amd.h): tiny operands (|x| ~ 1e-30, products
    ~ 1e-60 * K: far below fp32's range -> exact zeros or denormal noise, never garbage), operands around 2^-100 (low pieces
    subnormal in bf16: >= 16 significant bits) and a single Inf activation (non-finite out wherever the fp32 path gives a
    non-finite value, finite everywhere else)."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W = 1, 64, 64, 16, 16
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    for scale, rel in ((1.0, 3e-6), (2.0 ** -50, 3e-6), (2.0 ** -100, 2e-4)):
        x = torch.randn(B, Cin, H, W, generator=g) * scale
        ref = F.conv2d(x.double(), w.double(), None, padding=1)
        bld = Builder(torch.device("cuda"))
        xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        y = bld.conv(xa, bld.pack_conv(dev(w), None), Cout, winograd=mode)
        bld.finish()
        bld.tape.run()
        out = from_nhwc(y.buf.reshape(B, H, W, Cout), Cout).double().cpu()
        err = (out - ref).abs().max().item()
        assert err < rel * ref.abs().max().item(), (scale, err, ref.abs().max().item())
    x = torch.randn(B, Cin, H, W, generator=g)
    x[0, 3, 8, 8] = float("inf")
    ref = F.conv2d(x, w, None, padding=1)
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), None), Cout, winograd=mode)
    bld.finish()
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, H, W, Cout), Cout).cpu()
    bad_ref = ~torch.isfinite(ref)
    assert bad_ref.any() and (~torch.isfinite(out))[bad_ref].all()  # Inf / NaN where the reference is non-finite ...
    far = torch.ones_like(bad_ref)
    far[:, :, 5:12, 5:12] = False  # (the Winograd tiles touching the Inf pixel may turn NaN as a whole: transforms subtract Inf)
    assert torch.isfinite(out[far]).all() and max_err(out[far], ref[far]) < conv_tol(Cin, 3, True)


@pytest.mark.parametrize("wino", [False, True, "x3", "wx3", "h2", "wh2"])
def test_conv2d_is_deterministic_across_launches(az, wino):
    """Race screen for the LDS-exchange epilogues and the split-K combine: 12 launches of the same convolution (gate,
    residual, SiLU; one with split-K) must give bit-identical outputs."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(77)
    for (B, Cin, Cout, H, W, splitk) in ((2, 96, 160, 40, 36, 0), (1, 256, 192, 16, 16, 4)):
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        b = torch.randn(Cout, generator=g)
        gate, res = torch.randn(B, Cout, generator=g), torch.randn(B, Cout, H, W, generator=g)
        bld = Builder(torch.device("cuda"))
        xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        ra = Act(to_nhwc(dev(res)).reshape(-1), B, H, W, Cout, Cout, True)
        gd = dev(gate)  # must outlive the tape: the descriptor only holds its address
        y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, act=1, gate=gd, gate_bstride=Cout, res=ra, winograd=wino)
        if splitk:
            a = bld.tape.keep[-1]
            a.splitk = splitk
            bld._ws_need = max(bld._ws_need, splitk * B * H * W * y.cs)
            bld._ws_users.append(a)
        bld.finish()
        bld.tape.run()
        first = y.buf.clone()
        for _ in range(11):
            y.buf.fill_(float("nan"))
            bld.tape.run()
            assert torch.equal(y.buf, first)


def test_graph_capture_replay(az):
    from azula_amd.engine import StepGraph, Tape

    x = torch.ones(1024, device="cuda")
    s = torch.full((1,), 2.0, device="cuda")
    tape = Tape()
    tape.add("az_scale_f32", x.data_ptr(), x.data_ptr(), s.data_ptr(), 1024)
    tape.add("az_scale_f32", x.data_ptr(), x.data_ptr(), s.data_ptr(), 1024)
    tape.run()
    graph = StepGraph(tape, x.device)
    assert graph.num_nodes == 2
    for _ in range(3):
        graph.launch()
    torch.cuda.synchronize()
    assert (x == 4.0**4).all()


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,stride", CONV_CASES + [(2, 768, 2304, 16, 16, 1, 1), (1, 200, 72, 11, 9, 3, 1)])
def test_conv2d_half_operands(az, B, Cin, Cout, H, W, ks, stride, half):
    """az_conv2d_{bf16,f16}_f32: bf16 / f16 MFMA operands, fp32 accumulation.  Checked against an fp64 convolution of the
    SAME rounded operands (the only remaining difference is fp32 accumulation order), and against the unrounded fp32
    result with the operand-rounding bound."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(Cin * Cout + H + ks)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    bld = Builder(torch.device("cuda"), half=half)
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, (Cin + 3) // 4 * 4, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, stride=stride)
    assert bld.tape.ops[-1][2] == ("az_conv2d_bf16_f32" if half == torch.bfloat16 else "az_conv2d_f16_f32")
    bld.finish()
    bld.tape.run()
    out = from_nhwc(y.buf.reshape(B, y.H, y.W, y.cs), Cout)
    xr, wr = x.to(half).double(), w.to(half).double()
    exact = F.conv2d(xr, wr, b.double(), stride=stride, padding=ks // 2).float()
    assert max_err(out, exact) < conv_tol(Cin, ks), max_err(out, exact)
    ref = F.conv2d(x, w, b, stride=stride, padding=ks // 2)
    eps = 2.0**-8 if half == torch.bfloat16 else 2.0**-11
    assert max_err(out, ref) < 4 * eps * max(1.0, ref.abs().max().item())
    assert (y.buf.reshape(B, y.H, y.W, y.cs)[..., Cout:] == 0).all()


@pytest.mark.parametrize("wino", [False, True, "wx3", "wh2"])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("res_kind", ["none", "same", "up", "nobias"])
def test_conv2d_epilogue_forms(az, wino, act, gated, res_kind):
    """Every (activation, gate, residual) form of the fused epilogue: the NHWC store batch is compiled per combination
    (conv.hip `epilogue_batch_nhwc`), the rest goes through the generic path; order ((v + bias) -> act) * gate + res."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(100 * act + 10 * gated + len(res_kind))
    B, Cin, Cout, H, W = 2, 24, 64, 16, 18
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = None if res_kind == "nobias" else torch.randn(Cout, generator=g)
    gate = torch.randn(B, Cout, generator=g) if gated else None
    res = {"none": None, "nobias": None, "same": torch.randn(B, Cout, H, W, generator=g),
           "up": torch.randn(B, Cout, H // 2, W // 2, generator=g)}[res_kind]
    ref = F.conv2d(x, w, b, padding=1)
    ref = [ref, F.silu(ref), F.relu(ref), F.relu(ref) ** 2][act]
    if gated:
        ref = ref * gate[:, :, None, None]
    if res is not None:
        ref = ref + (res if res_kind == "same" else F.interpolate(res, scale_factor=(2.0, 2.0), mode="nearest"))
    bld = Builder(torch.device("cuda"))
    xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
    ra = None
    if res is not None:
        ra = Act(to_nhwc(dev(res)).reshape(-1), B, res.shape[2], res.shape[3], Cout, Cout, True)
    y = bld.conv(xa, bld.pack_conv(dev(w), dev(b) if b is not None else None), Cout, act=act,
                 gate=dev(gate) if gated else None, gate_bstride=Cout, res=ra, res_up=int(res_kind == "up"), winograd=wino)
    bld.finish()
    assert [n for _, _, n in bld.tape.ops if n.startswith("az_conv2d")] == [WINO_NAME[wino]]
    bld.tape.run()
    err = max_err(from_nhwc(y.buf.reshape(B, H, W, Cout), Cout), ref)
    assert err < conv_tol(Cin, 3, wino) * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize(
    "case",
    [  # (B, Cin, Cout, H, W, ks, stride, winograd, form)
        (2, 256, 128, 8, 8, 3, 1, False, "plain"),
        (2, 256, 128, 8, 8, 3, 1, False, "gate_res"),
        (4, 256, 256, 16, 16, 3, 1, True, "silu"),
        (4, 512, 256, 8, 8, 3, 1, True, "gate_res"),
        (4, 256, 256, 16, 16, 3, 1, "wx3", "silu"),
        (4, 512, 256, 8, 8, 3, 1, "wx3", "gate_res"),
        (1, 1024, 96, 6, 6, 1, 1, False, "plain"),       # 24 quads (not a divisor of 256), 36 pixels in 2 ragged chunks
        (2, 128, 1280, 8, 8, 3, 2, False, "plain"),      # 320 quads: a thread owns two quads in turn; stride 2
        (2, 256, 128, 8, 8, 3, 1, False, "concat"),
    ],
)
def test_groupnorm_statistics_from_the_splitk_combine(az, case, monkeypatch):
    """AzConvArgs.gn_quads with splitk > 1 (the small maps): the combine kernel leaves (n, mean, M2) per (image, pixel chunk,
    channel quad) and the following GroupNorm skips its statistics pass.  Same bar as the Winograd-epilogue test below."""
    from azula_amd import engine
    from azula_amd.engine import Act, Builder

    B, Cin, Cout, H, W, ks, stride, wino, form = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H * stride, W * stride, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(ks * ks * Cin)
    b = torch.randn(Cout, generator=g) + 30.0
    gate = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g) * 0.5 + 50.0
    conv = F.conv2d(x, w, b, padding=ks // 2, stride=stride)
    assert conv.shape[-2:] == (H, W)
    if form in ("gate_res", "concat"):
        conv = conv * gate[:, :, None, None] + res
    elif form == "silu":
        conv = F.silu(conv)
    w2 = torch.randn(64, Cin, ks, ks, generator=g) / math.sqrt(ks * ks * Cin)
    two = form == "concat"
    full = torch.cat((conv, F.conv2d(x, w2, None, padding=ks // 2, stride=stride)), 1) if two else conv
    C2, groups = full.shape[1], 6 if two else 8  # (a group never straddles the two sources)
    gw, gb = torch.randn(C2, generator=g), torch.randn(C2, generator=g)
    ref = F.silu(F.group_norm(full, groups, gw, gb, eps=1e-5))

    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(engine, "GN_FUSED", fused)
        bld = Builder(torch.device("cuda"))
        xa = Act(to_nhwc(dev(x)).reshape(-1), B, H * stride, W * stride, Cin, Cin, True)
        ra = Act(to_nhwc(dev(res)).reshape(-1), B, H, W, Cout, Cout, True)
        kw = dict(gate=dev(gate), gate_bstride=Cout, res=ra) if form in ("gate_res", "concat") else {}
        y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, act=int(form == "silu"), stride=stride, winograd=wino, gn_stats=True, **kw)
        y1 = bld.conv(xa, bld.pack_conv(dev(w2), None), 64, stride=stride, winograd=wino, gn_stats=True) if two else None
        n = bld.group_norm(y, groups, weight=dev(gw), bias=dev(gb), act=1, x1=y1)
        bld.finish()
        convs = [args[0]._obj for _, args, nm in bld.tape.ops if nm.startswith("az_conv2d")]
        assert all(c.splitk > 1 for c in convs), [c.splitk for c in convs]
        names = [nm for _, _, nm in bld.tape.ops]
        assert (y.gn_quads is not None) == fused
        assert ("az_groupnorm_stats_f32" in names) == (not fused), names
        bld.tape.run()
        outs[fused] = from_nhwc(n.buf.reshape(B, H, W, C2), C2).clone()
        bld.tape.run()
        assert torch.equal(from_nhwc(n.buf.reshape(B, H, W, C2), C2), outs[fused]), "not deterministic"
    e_ref, e_ab = max_err(outs[True], ref), max_err(outs[True], outs[False])
    print(f"{case}: fused vs torch {e_ref:.2e}, fused vs separate pass {e_ab:.2e}")
    assert e_ref < 6e-5 and max_err(outs[False], ref) < 6e-5
    assert e_ab < 2.5e-5


@pytest.mark.parametrize("mode", [True, "wx3", "wh2"])
@pytest.mark.parametrize("form", ["plain", "gate_res", "silu", "concat", "mixed"])
def test_groupnorm_statistics_from_the_conv_epilogue(az, form, mode, monkeypatch):
    """AzConvArgs.gn_quads: the Winograd epilogue leaves (n, mean, M2) per (image, 64-tile block, channel quad) and the
    following GroupNorm skips its statistics pass.  Checked against torch's group_norm of the torch conv, with a large
    common offset (bias 30, residual mean 50: mean >> std, the case a naive sum-of-squares loses) and against the
    separate statistics pass on the same convolution output (AZ_GN_FUSED = 0)."""
    from azula_amd import engine
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(len(form))
    B, Cin, Cout, H, W, groups = 2, 16, 128, 32, 16, 32  # 16 x 8 tiles = 2 tile blocks per image
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g) + 30.0
    gate = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g) * 0.5 + 50.0
    gw, gb = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    conv = F.conv2d(x, w, b, padding=1)
    if form in ("gate_res", "concat", "mixed"):
        conv = conv * gate[:, :, None, None] + res
    elif form == "silu":
        conv = F.silu(conv)
    w2 = torch.randn(128, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    conv2 = F.conv2d(x, w2, None, padding=1)
    two = form in ("concat", "mixed")
    full = torch.cat((conv, conv2), 1) if two else conv
    C2 = full.shape[1]
    gw, gb = torch.randn(C2, generator=g), torch.randn(C2, generator=g)
    ref = F.silu(F.group_norm(full, groups, gw, gb, eps=1e-5))

    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(engine, "GN_FUSED", fused)
        bld = Builder(torch.device("cuda"))
        xa = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        ra = Act(to_nhwc(dev(res)).reshape(-1), B, H, W, Cout, Cout, True)
        kw = dict(gate=dev(gate), gate_bstride=Cout, res=ra) if form in ("gate_res", "concat", "mixed") else {}
        y = bld.conv(xa, bld.pack_conv(dev(w), dev(b)), Cout, act=int(form == "silu"), winograd=mode, gn_stats=True, **kw)
        y1 = None
        if two:  # "mixed": the second source carries no moments -> the whole norm falls back to the statistics pass
            y1 = bld.conv(xa, bld.pack_conv(dev(w2), None), 128, winograd=mode, gn_stats=form == "concat")
        n = bld.group_norm(y, groups, weight=dev(gw), bias=dev(gb), act=1, x1=y1)
        bld.finish()
        names = [nm for _, _, nm in bld.tape.ops]
        assert (y.gn_quads is not None) == fused
        assert ("az_groupnorm_stats_f32" in names) == (not fused or form == "mixed"), names
        bld.tape.run()
        outs[fused] = from_nhwc(n.buf.reshape(B, H, W, C2), C2).clone()
        bld.tape.run()
        assert torch.equal(from_nhwc(n.buf.reshape(B, H, W, C2), C2), outs[fused]), "not deterministic"
    e_ref, e_ab = max_err(outs[True], ref), max_err(outs[True], outs[False])
    print(f"{form}: fused vs torch {e_ref:.2e}, fused vs separate pass {e_ab:.2e}")
    assert e_ref < 6e-5 and max_err(outs[False], ref) < 6e-5  # measured 2.9e-6 .. 1.3e-5 (O(1) outputs from data with mean/std up to 100)
    assert e_ab < 2.5e-5  # measured <= 4.8e-6


def test_calibration_kernels(az):
    """Measurement support of bench.py / tools/pmc_traffic.py (never on the sampling path): the known-traffic kernels touch
    what they say and the matrix-pipe kernel sustains a plausible fp32 MFMA rate (the guide's peak is 157.3 TF/s)."""
    n = 1 << 22
    src = torch.arange(n, device="cuda", dtype=torch.float32)
    sink = torch.zeros(64, device="cuda")
    dst = torch.empty(n, device="cuda")
    az.call("az_calib_write_f32", dst.data_ptr(), 4 * n, 2.5, az.stream_ptr())
    assert (dst == 2.5).all()
    az.call("az_calib_read_f32", src.data_ptr(), sink.data_ptr(), 4 * n, 16, 64, 4096, az.stream_ptr())
    wgs, iters = 512, 2000
    for _ in range(2):
        az.call("az_calib_mfma_f32", sink.data_ptr(), wgs, iters, 1.0, 0.5, az.stream_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    az.call("az_calib_mfma_f32", sink.data_ptr(), wgs, iters, 1.0, 0.5, az.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    tf = wgs * 4 * iters * 8 * 4096 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"az_calib_mfma_f32: {tf:.1f} TF/s")
    assert 20.0 < tf < 170.0  # (a loose sanity bound: a cold clock has shown 74 TF/s on a first launch)
    e0.record()
    az.call("az_calib_mfma_random_f32", sink.data_ptr(), wgs, iters, 1.0, 0.5, az.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    tf = wgs * 4 * iters * 8 * 4096 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"az_calib_mfma_random_f32: {tf:.1f} TF/s")
    assert 20.0 < tf < 170.0
    e0.record()
    az.call("az_calib_mfma_random_bf16", sink.data_ptr(), wgs, 2 * iters, 1.0, 0.5, az.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    tf = wgs * 4 * 2 * iters * 8 * 32768 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"az_calib_mfma_random_bf16: {tf:.1f} TF/s")
    assert 300.0 < tf < 2700.0  # (nominal 2516.8; ~1700 - 1800 under the 1400 W cap)



@pytest.mark.parametrize("shape", [(2, 512, 768, 768), (1, 300, 64, 260), (3, 130, 128, 388), (1, 4096, 3072, 768), (1, 1000, 256, 1024),
                                   (64, 256, 128, 2304), (16, 1024, 3072, 768)])
@pytest.mark.parametrize("act,res", [(0, False), (1, True)])
@pytest.mark.parametrize("mode", ["x3", "h2"])  # ("h2": the f16x2 form of the same kernels -- two half pieces, three products)
def test_x3_gemm_big_tile(az, monkeypatch, shape, act, res, mode):
    """conv_gemm_x3_big_kernel (256 x 256 tile, 8 waves, two LDS stages) against the 128 x 128 bf16x3 kernel and fp64: same six
    partial products per K step of 16 channels, so the two agree to a few ulps; ragged token / channel tiles, split-K, residual."""
    from azula_amd.engine import Act, Builder

    B, T, Cin, Cout = shape
    g = torch.Generator().manual_seed(T + Cout)
    x = torch.randn(B, Cin, T, 1, generator=g) * torch.logspace(-2, 2, Cin)[None, :, None, None]
    w = torch.randn(Cout, Cin, generator=g) / Cin**0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, T, 1, generator=g) if res else None
    ref = F.conv2d(x.double(), w.double()[:, :, None, None], b.double())
    if act:
        ref = F.silu(ref)
    if res:
        ref = ref + r.double()
    outs = {}
    for big in ("0", "1", "3", "plan"):  # 128 x 128 everywhere / 256 x 256 wherever eligible / 192-cout tiles / the library's own plan
        if big == "plan":
            monkeypatch.delenv("AZ_X3_BIG")
        else:
            monkeypatch.setenv("AZ_X3_BIG", big)
            monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        bld = Builder(torch.device("cuda"))
        xin = Act(to_nhwc(dev(x)).reshape(-1), B, T, 1, Cin, Cin, True)  # (kept alive: the tape holds raw addresses)
        rin = Act(to_nhwc(dev(r)).reshape(-1), B, T, 1, Cout, (Cout + 3) // 4 * 4, True) if res else None
        y = bld.conv(xin, bld.pack_conv(dev(w), dev(b)), Cout, act=act, res=rin, winograd=mode)
        bld.finish()
        bld.tape.run()
        outs[big] = from_nhwc(y.buf.reshape(B, T, 1, -1), Cout).double().cpu()
    scale = ref.abs().max().item()
    e0, e1 = (outs["0"] - ref).abs().max().item() / scale, (outs["1"] - ref).abs().max().item() / scale
    e2, e3 = (outs["plan"] - ref).abs().max().item() / scale, (outs["3"] - ref).abs().max().item() / scale
    print(shape, act, "128 tile", e0, "256 tile", e1, "192-cout tile", e3, "plan", e2, "between", (outs["1"] - outs["0"]).abs().max().item() / scale)
    assert e1 < 2e-6 and e0 < 2e-6 and e2 < 2e-6 and e3 < 2e-6


@pytest.mark.parametrize("shape", [(4, 64, 64, 256, 256, 256), (1, 96, 100, 64, 128, 320), (2, 48, 48, 512, 256, 256)])
@pytest.mark.parametrize("mode", ["x3", "h2"])
def test_x3_gemm_big_tile_two_sources(az, monkeypatch, shape, mode):
    """The 256 x 256 bf16x3 kernel on a channel concatenation read in place (the 1x1 skip convolutions of ADM's decoder,
    plugins/adm/_src/unet.py:215,631): K steps walk source 0, then source 1; against the 128 x 128 kernel and fp64."""
    from azula_amd.engine import Act, Builder

    B, H, W, C0, C1, Cout = shape
    g = torch.Generator().manual_seed(H + Cout)
    x0, x1 = torch.randn(B, C0, H, W, generator=g), torch.randn(B, C1, H, W, generator=g) * 3
    w = torch.randn(Cout, C0 + C1, generator=g) / (C0 + C1) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(torch.cat([x0, x1], 1).double(), w.double()[:, :, None, None], b.double())
    outs = {}
    for big in ("0", "1"):
        monkeypatch.setenv("AZ_X3_BIG", big)
        monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        bld = Builder(torch.device("cuda"))
        a0 = Act(to_nhwc(dev(x0)).reshape(-1), B, H, W, C0, C0, True)
        a1 = Act(to_nhwc(dev(x1)).reshape(-1), B, H, W, C1, C1, True)
        y = bld.conv(a0, bld.pack_conv(dev(w), dev(b), cin0=C0), Cout, src1=a1, winograd=mode)
        bld.finish()
        bld.tape.run()
        outs[big] = from_nhwc(y.buf.reshape(B, H, W, -1), Cout).double().cpu()
    scale = ref.abs().max().item()
    e0, e1 = (outs["0"] - ref).abs().max().item() / scale, (outs["1"] - ref).abs().max().item() / scale
    print(shape, "128 tile", e0, "256 tile", e1, "between", (outs["1"] - outs["0"]).abs().max().item() / scale)
    assert e1 < 2e-6 and e0 < 2e-6


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 512, 768, 768), (1, 300, 64, 260), (3, 130, 128, 388), (1, 4096, 3072, 768), (64, 256, 128, 2304)])
def test_half_gemm_big_tile(az, monkeypatch, shape, half):
    """conv_gemm_half_big_kernel (256 x 256 tile for modules cast to half precision) against the 128 x 128 half kernel (same
    rounding of the operands) and an fp64 product of the rounded operands."""
    from azula_amd.engine import Act, Builder

    B, T, Cin, Cout = shape
    g = torch.Generator().manual_seed(T + Cout)
    x = torch.randn(B, Cin, T, 1, generator=g)
    w = torch.randn(Cout, Cin, generator=g) / Cin**0.5
    b = torch.randn(Cout, generator=g)
    exact = F.conv2d(x.to(half).double(), w.to(half).double()[:, :, None, None], b.double())
    outs = {}
    for big in ("0", "1", "plan"):
        if big == "plan":
            monkeypatch.delenv("AZ_X3_BIG")
        else:
            monkeypatch.setenv("AZ_X3_BIG", big)
            monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        bld = Builder(torch.device("cuda"), half=half)
        xin = Act(to_nhwc(dev(x)).reshape(-1), B, T, 1, Cin, Cin, True)
        y = bld.conv(xin, bld.pack_conv(dev(w), dev(b)), Cout, act=1)
        bld.finish()
        bld.tape.run()
        outs[big] = from_nhwc(y.buf.reshape(B, T, 1, -1), Cout).double().cpu()
    ref = F.silu(exact)
    for k, o in outs.items():
        assert (o - ref).abs().max().item() < conv_tol(Cin, 1), (k, (o - ref).abs().max().item())
    print(shape, half, "128 vs 256 tile", (outs["1"] - outs["0"]).abs().max().item(), "plan", (outs["plan"] - outs["0"]).abs().max().item())
    # (equal to the last bit where the 128 x 128 launch is not split along K; a split changes the summation order)
    assert (outs["1"] - outs["0"]).abs().max().item() < 2e-5 and (outs["plan"] - outs["0"]).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,stride", [(2, 64, 256, 40, 36, 3, 2), (1, 32, 320, 33, 47, 3, 1), (3, 48, 256, 17, 19, 5, 2),
                                                      (2, 128, 256, 64, 64, 3, 2), (1, 64, 200, 30, 30, 1, 2), (2, 32, 256, 21, 21, 7, 3)])
@pytest.mark.parametrize("mode", ["x3", "h2"])
def test_x3_big_tile_with_taps(az, monkeypatch, B, Cin, Cout, H, W, ks, stride, mode):
    """conv_gemm_x3_big_kernel<4, TAPS>: k x k filters, strides, zero padding, tiles that span image borders and several images --
    against the 128 x 128 bf16x3 kernel (same products; the K walk is tap-major in both) and an fp64 convolution."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(H * W + ks)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=ks // 2))
    outs = {}
    for big in ("0", "1"):
        monkeypatch.setenv("AZ_X3_BIG", big)
        monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        bld = Builder(torch.device("cuda"))
        xin = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        y = bld.conv(xin, bld.pack_conv(dev(w), dev(b)), Cout, stride=stride, act=1, winograd=mode)
        bld.finish()
        bld.tape.run()
        outs[big] = from_nhwc(y.buf.reshape(B, y.H, y.W, -1), Cout).double().cpu()
    e0, e1 = (outs["0"] - ref).abs().max().item(), (outs["1"] - ref).abs().max().item()
    print((B, Cin, Cout, H, W, ks, stride), "128 tile", e0, "256 tile", e1, "between", (outs["1"] - outs["0"]).abs().max().item())
    assert e0 < conv_tol(Cin, ks, "x3") and e1 < conv_tol(Cin, ks, "x3")


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,stride", [(2, 64, 256, 40, 36, 3, 2), (1, 128, 320, 33, 47, 3, 1), (2, 64, 256, 17, 19, 5, 2),
                                                      (1, 64, 200, 30, 30, 1, 2)])
def test_half_big_tile_with_taps(az, monkeypatch, B, Cin, Cout, H, W, ks, stride, half):
    """conv_gemm_half_big_kernel<F16, TAPS>: k x k filters, strides, zero padding on the 256 x 256 tile of half-precision modules,
    against the 128 x 128 half kernel and an fp64 convolution of the rounded operands."""
    from azula_amd.engine import Act, Builder

    g = torch.Generator().manual_seed(H * W + ks)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    exact = F.conv2d(x.to(half).double(), w.to(half).double(), b.double(), stride=stride, padding=ks // 2)
    outs = {}
    for big in ("0", "1"):
        monkeypatch.setenv("AZ_X3_BIG", big)
        monkeypatch.setenv("AZ_DEBUG_AB", "1")  # (A/B overrides are honoured only under the debug switch)
        bld = Builder(torch.device("cuda"), half=half)
        xin = Act(to_nhwc(dev(x)).reshape(-1), B, H, W, Cin, Cin, True)
        y = bld.conv(xin, bld.pack_conv(dev(w), dev(b)), Cout, stride=stride)
        bld.finish()
        bld.tape.run()
        outs[big] = from_nhwc(y.buf.reshape(B, y.H, y.W, -1), Cout).double().cpu()
    e0, e1 = (outs["0"] - exact).abs().max().item(), (outs["1"] - exact).abs().max().item()
    print((B, Cin, Cout, H, W, ks, stride), half, "128 tile", e0, "256 tile", e1)
    assert e0 < conv_tol(Cin, ks) and e1 < conv_tol(Cin, ks)
