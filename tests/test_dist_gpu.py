r"""N > 1 on the GPU, rehearsed on ONE device: two processes (gloo rendezvous on 127.0.0.1) share cuda:0, each samples
its shard through the fused hipGraph loop with the device RNG, and the all-gathered x0 equals the single-process run
sample for sample up to fp32 round-off (every rank draws the full-batch noise and keeps its slice).  On an 8-GPU node the same code runs
one process per GPU over RCCL (``bench.py --gpus N``); only the collective backend differs."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def build():
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated, UNet
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDPMSampler

    torch.manual_seed(0)
    net = TimeModulated(UNet(3, 3, hid_channels=(16, 32), hid_blocks=(1, 1), norm="group", groups=4, mod_features=16), 16, name="unet")
    for p in net.parameters():
        p.data.normal_(std=0.2)
    den = KarrasDenoiser(net, VPSchedule()).cuda().eval()
    return DDPMSampler(den, steps=6, silent=True)


def worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_grad_enabled(False)
        from azula_amd.parallel import init_sharded, sample_sharded

        smp = build()
        torch.manual_seed(1)
        x_local = init_sharded(smp, (4, 3, 16, 16), device="cuda")
        torch.manual_seed(2)
        x0 = sample_sharded(smp, x_local)
        assert x0.shape == (4, 3, 16, 16) and next(iter(smp._fused_cache.values())).graph is not None
        if rank == 0:
            torch.save(x0.cpu(), out_path)
    finally:
        dist.destroy_process_group()


def test_two_processes_on_one_gpu_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "x0.pt")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    sharded = torch.load(out)
    torch.set_grad_enabled(False)
    smp = build()
    torch.manual_seed(1)
    x1 = smp.init((4, 3, 16, 16), device="cuda")
    torch.manual_seed(2)
    single = smp(x1).cpu()
    # same noise sample for sample; the shard runs batch-2 plans (other split-K / tile choices than batch 4), hence
    # fp32 round-off instead of bit equality
    scale = max(1.0, single.abs().max().item())
    assert (sharded - single).abs().max().item() < 5e-5 * scale


def rccl_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" IS RCCL on ROCm
    try:
        torch.set_grad_enabled(False)
        from azula_amd.parallel import init_sharded, sample_sharded

        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        smp = build()
        torch.manual_seed(1)
        x_local = init_sharded(smp, (4, 3, 16, 16), device="cuda")
        torch.manual_seed(2)
        timings = {}
        x0 = sample_sharded(smp, x_local, timings=timings)  # fused loop, then dist.all_gather_into_tensor over RCCL
        assert x0.shape == (4, 3, 16, 16) and timings["allgather_ms"] > 0
        # an all-reduce too: the collective bench.py uses for its max-over-ranks clock
        t = torch.tensor([3.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == 3.0
        torch.save(x0.cpu(), out_path)
    finally:
        dist.destroy_process_group()


def test_rccl_backend_world_of_one(tmp_path):
    """RCCL itself: a process group on the `nccl` backend with ONE rank on the GPU box.  Proves that librccl loads, a
    communicator initialises on this device and `sample_sharded`'s all_gather_into_tensor path executes; the 8-GPU run
    differs only in the world size (the driver launches it at round end)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "x0.pt")
    mp.spawn(rccl_worker, args=(1, port, out), nprocs=1, join=True)
    gathered = torch.load(out)
    torch.set_grad_enabled(False)
    smp = build()
    torch.manual_seed(1)
    x1 = smp.init((4, 3, 16, 16), device="cuda")
    torch.manual_seed(2)
    assert torch.equal(gathered, smp(x1).cpu())


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE in the environment (the form the driver uses for
    N = 1) must start its own ranks and print ONE JSON line; here 2 ranks share the box's GPU over gloo."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["AZ_DIST_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "2",
                          "--warmup", "1", "--no-pmc", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist"]["ranks_seen"] == 2 and out["dist"]["world_size"] == 2
    assert out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"] and out["value"] > 0
