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
