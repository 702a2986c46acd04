r"""N > 1 path on CPU: two gloo processes, batch sharded, one all-gather of x0, and the result equals
the single-process run sample for sample (global-noise slicing)."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l1, self.l2 = torch.nn.Linear(5, 32), torch.nn.Linear(32, 5)

    def forward(self, x, t, **kw):
        return self.l2(torch.tanh(self.l1(x) + t))


def make_sampler(kind, steps):
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler, DDPMSampler

    torch.manual_seed(0)
    den = KarrasDenoiser(Toy(), VPSchedule()).eval()
    return DDPMSampler(den, steps=steps, silent=True) if kind == "ddpm" else DDIMSampler(den, eta=0.3, steps=steps, silent=True)


def worker(rank, world, port, kind, out_path, batch=8):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from azula_amd.parallel import init_sharded, sample_sharded, shard_range

        smp = make_sampler(kind, 16)
        torch.manual_seed(1)
        x_local = init_sharded(smp, (batch, 5))
        assert x_local.shape == (batch // world, 5)
        assert shard_range(batch, rank, world) == range(rank * batch // world, (rank + 1) * batch // world)
        torch.manual_seed(2)
        timings = {}
        x0 = sample_sharded(smp, x_local, timings=timings)
        assert x0.shape == (batch, 5) and timings["sample_ms"] > 0 and timings["allgather_ms"] > 0
        # every rank holds the same gathered tensor
        ref = [torch.empty_like(x0) for _ in range(world)]
        dist.all_gather(ref, x0)
        assert all(torch.equal(r, x0) for r in ref)
        if rank == 0:
            torch.save(x0, out_path)
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind", ["ddpm", "ddim"])
def test_two_rank_sharded_sampling_equals_single_process(tmp_path, kind):
    out = str(tmp_path / "x0.pt")
    mp.spawn(worker, args=(2, free_port(), kind, out), nprocs=2, join=True)
    x0 = torch.load(out)
    smp = make_sampler(kind, 16)
    torch.manual_seed(1)
    x1 = smp.init((8, 5))
    torch.manual_seed(2)
    ref = smp(x1)
    assert torch.equal(x0, ref)


@pytest.mark.parametrize("world", [4, 8])
def test_c4_shaped_sharding_256_images(tmp_path, world):
    """BASELINE configs[3]'s shard arithmetic: a global batch of 256 DDPM trajectories split 8 x 32 (and 4 x 64); every
    rank draws the FULL-batch noise of each step and keeps its slice, so the gathered x0 is bit-equal to one process."""
    out = str(tmp_path / "x0.pt")
    mp.spawn(worker, args=(world, free_port(), "ddpm", out, 256), nprocs=world, join=True)
    x0 = torch.load(out)
    smp = make_sampler("ddpm", 16)
    torch.manual_seed(1)
    x1 = smp.init((256, 5))
    torch.manual_seed(2)
    ref = smp(x1)
    assert torch.equal(x0, ref)


def test_shard_range():
    from azula_amd.parallel import shard_range

    assert list(shard_range(8, 1, 4)) == [2, 3]
    with pytest.raises(ValueError):
        shard_range(6, 0, 4)
