r"""API usages off the benchmark's beaten path, on the GPU: per-sample times, an fp64 sampler clock, start / stop other than
(1, 0), non-contiguous latents, every schedule x denoiser x sampler family combination through the fused or generic loop,
plan-cache reuse across batch sizes, ADM with per-sample times and a tensor-valued guidance strength.  Shape / finiteness /
consistency checks (the numerical parity of each component is covered elsewhere)."""

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_unusual_but_valid_usages():
    from azula_amd.denoise import KarrasDenoiser, SimpleDenoiser
    from azula_amd.nn import TimeModulated, UNet
    from azula_amd.noise import VPSchedule, CosineSchedule, RectifiedSchedule
    from azula_amd.sample import DDIMSampler, DDPMSampler, EulerSampler, HeunSampler, zEABSampler, PCSampler
    from azula_amd.guidance import CFGDenoiser
    from azula_amd.plugins import adm

    net = TimeModulated(UNet(3, 3, hid_channels=(16, 32), hid_blocks=(1, 1), norm="group", groups=4, mod_features=16), 16, name="unet")
    den = KarrasDenoiser(net, VPSchedule()).cuda().eval()
    x = torch.randn(3, 3, 24, 20, device="cuda")
    # 1. per-sample times
    q = den(x, torch.rand(3, device="cuda")); assert q.mean.shape == x.shape and torch.isfinite(q.mean).all()
    # 2. fp64 sampler clock, fp32 latents
    s = DDIMSampler(den, steps=5, silent=True, dtype=torch.float64)
    x0 = s(s.init(x.shape, device="cuda").float()); assert x0.dtype == torch.float64 and torch.isfinite(x0).all()  # promoted like the reference (G11)
    # 3. start/stop other than (1, 0); non-contiguous input
    s = DDPMSampler(den, start=0.8, stop=0.1, steps=4, silent=True)
    x0 = s(x.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)); assert torch.isfinite(x0).all()
    # 4. other schedules through the fused loop and SimpleDenoiser
    for sch in (CosineSchedule(), RectifiedSchedule()):
        for D in (KarrasDenoiser, SimpleDenoiser):
            d = D(net, sch).cuda().eval()
            for S in (DDIMSampler, EulerSampler, HeunSampler, zEABSampler, PCSampler):
                smp = S(d, steps=4, silent=True)
                o = smp(smp.init(x.shape, device="cuda")); assert torch.isfinite(o).all(), (sch, D, S)
    # 5. same sampler object, two batch sizes (plan cache), then back
    smp = DDIMSampler(den, steps=3, silent=True)
    a = smp(x); b = smp(x[:2]); c = smp(x); assert torch.equal(a, c) and torch.allclose(a[:2], b, atol=1e-5)
    # 6. ADM with per-sample float times via the denoiser, guidance as a tensor
    ad = adm.make_model(image_size=32, num_channels=32, num_res_blocks=1, channel_mult=(1, 2), attention_resolutions=(16,), num_heads=2,
                        num_head_channels=-1, num_classes=5, learn_var=True, clip_mean=True, resblock_updown=True, use_scale_shift_norm=True).cuda().eval()
    xa = torch.randn(2, 3, 32, 32, device="cuda"); lab = torch.tensor([1, 3], device="cuda")
    q = ad(xa, torch.tensor([0.3, 0.9], device="cuda"), label=lab); assert torch.isfinite(q.mean).all() and q.var.shape == q.mean.shape
    g = CFGDenoiser(ad)
    o = DDIMSampler(g, steps=3, silent=True)(xa, positive={"label": lab}, negative={"label": torch.zeros_like(lab)}, guidance=torch.tensor(1.5))
    assert torch.isfinite(o).all()
