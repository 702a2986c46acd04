r"""``az_randn_slice_f32``: a rank's slice of the full-batch noise, bit-identical to the same elements of ``torch.randn`` on the
device (N GPUs reproduce the single-device random stream; the reference draws ``randn_like(x_t)`` of the whole batch,
azula/sample.py:214,259)."""

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,world", [((8, 3, 64, 64), 2), ((8, 3, 256, 256), 8), ((256, 3, 256, 256), 8), ((6, 5, 7, 9), 3), ((4, 1000), 4)])
def test_sliced_draw_equals_the_full_draw(shape, world):
    from azula_amd.sample import DDPMSampler

    torch.cuda.init()
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    per = shape[0] // world
    like = torch.empty((per, *shape[1:]), device="cuda")
    smp = DDPMSampler.__new__(DDPMSampler)
    for seed in (0, 1234):
        torch.manual_seed(seed)
        _ = torch.randn(1000, device="cuda")  # (a non-zero Philox offset)
        start = gen.get_offset()
        full = torch.randn(shape, device="cuda")
        full2 = torch.randn(shape, device="cuda")  # the NEXT draw: the offset must have advanced as the full draw advances it
        end = gen.get_offset()
        for rank in range(world):
            torch.manual_seed(seed)
            _ = torch.randn(1000, device="cuda")
            assert gen.get_offset() == start
            smp.shard = (rank, world)
            mine = smp._draw_noise(like)
            assert torch.equal(mine, full[rank * per : (rank + 1) * per]), (shape, world, rank)
            buf = torch.empty_like(like)
            assert smp._draw_noise(buf, out=buf) is buf and torch.equal(buf, full2[rank * per : (rank + 1) * per])
            assert gen.get_offset() == end


def test_sliced_draw_in_a_sharded_ddpm_run(golden):
    """DDPM-8 of the small golden UNet: each of 2 'ranks' (run one after the other on this GPU) with the sliced draw reproduces
    its half of the single-device run (the same noise bit for bit -- the test above; the backbone's tile plans may differ with
    the batch, hence a round-off bound on the trajectories)."""
    from test_gpu_fp64 import unet_denoiser
    from azula_amd.sample import DDPMSampler

    g = golden("g11_sampler_dtype")
    den = unet_denoiser(g)
    x1 = g["unet_x1"].cuda()
    B = x1.shape[0] - x1.shape[0] % 2
    x1 = x1[:B].contiguous()
    torch.manual_seed(7)
    ref = DDPMSampler(den, steps=8, silent=True)(x1)
    for rank in range(2):
        smp = DDPMSampler(den, steps=8, silent=True)
        smp.shard = (rank, 2)
        torch.manual_seed(7)
        out = smp(x1[rank * B // 2 : (rank + 1) * B // 2].contiguous())
        e = (out - ref[rank * B // 2 : (rank + 1) * B // 2]).abs().max().item()
        print("rank", rank, "sliced-noise DDPM-8 vs the single-device run: max|d|", e)
        assert e < 1e-4 * max(1.0, ref.abs().max().item())


def test_sliced_draw_self_check_and_fallback(monkeypatch):
    """VERDICT r05 weak #7 / ADVICE: the slice kernel re-derives ATen's launch policy, so the first sharded draw on a device
    verifies it against ``torch.randn`` (small tensor, multi-iteration tensor, generator offset) without disturbing the
    generator; when the check fails (simulated here) the draw-and-slice path gives the same elements."""
    from azula_amd import sample
    from azula_amd.sample import DDPMSampler

    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    torch.manual_seed(99)
    _ = torch.randn(10, device="cuda")
    before = (gen.initial_seed(), gen.get_offset())
    sample._SLICE_VERIFIED.clear()
    assert sample._randn_slice_verified(torch.device("cuda", torch.cuda.current_device())) is True
    assert (gen.initial_seed(), gen.get_offset()) == before  # (the check restores the generator)
    smp = DDPMSampler.__new__(DDPMSampler)
    smp.shard = (1, 2)
    like = torch.empty(3, 5, 7, device="cuda")
    torch.manual_seed(5)
    a = smp._draw_noise(like)
    off_a = gen.get_offset()
    monkeypatch.setattr(sample, "_randn_slice_verified", lambda dev: False)
    with monkeypatch.context() as m:
        m.setattr(sample, "_randn_slice", lambda *a, **k: (_ for _ in ()).throw(AssertionError("slice kernel used after a failed check")))
        torch.manual_seed(5)
        b = smp._draw_noise(like)
    assert torch.equal(a, b) and gen.get_offset() == off_a


@pytest.mark.parametrize("device", ["cuda", "cpu"])
def test_init_sharded_forms_only_the_local_rows(golden, device, monkeypatch):
    """`init_sharded`: the rank's rows of `Sampler.init` of the full batch, bit for bit, without the
    full-batch tensor -- scalar / broadcast mean and var through the sliced draw, per-sample tensors through draw-and-slice."""
    from test_gpu_fp64 import unet_denoiser
    from azula_amd import parallel
    from azula_amd.sample import DDIMSampler

    den = unet_denoiser(golden("g11_sampler_dtype"))
    smp = DDIMSampler(den, steps=4, silent=True)
    shape = (8, 3, 16, 16)
    for kw in ({}, {"mean": 0.3, "var": 2.0}, {"mean": torch.linspace(-1, 1, 3, device=device).reshape(1, 3, 1, 1), "var": 0.5},
               {"mean": torch.linspace(-1, 1, 8, device=device).reshape(8, 1, 1, 1).expand(shape).contiguous()}):
        torch.manual_seed(3)
        full = smp.init(shape, device=device, **kw)
        end = torch.cuda.default_generators[torch.cuda.current_device()].get_offset() if device == "cuda" else None
        for world in (2, 4):
            for rank in range(world):
                monkeypatch.setattr(parallel, "_world", lambda group=None, r=rank, w=world: (r, w))
                torch.manual_seed(3)
                mine = parallel.init_sharded(smp, shape, device=device, **kw)
                per = shape[0] // world
                assert mine.shape == (per, *shape[1:]) and mine.device.type == device
                assert torch.equal(mine, full[rank * per : (rank + 1) * per]), (kw.keys(), world, rank)
                if end is not None:
                    assert torch.cuda.default_generators[torch.cuda.current_device()].get_offset() == end
