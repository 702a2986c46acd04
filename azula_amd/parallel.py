r"""Batch-parallel sampling over the GPUs of one node.

Every sample's reverse trajectory is independent (no cross-sample op exists anywhere on the
path), so the batch is sharded across one process per GPU with replicated weights and NO
collective inside the loop; the only exchange is ONE all-gather of the final ``x0`` (RCCL over
xGMI with the ``nccl`` backend; ``gloo`` on CPU for the tests).  The reference has no
distributed code at all -- this module is new.

Determinism: every rank seeds its generator identically and takes ITS slice of the full-batch noise
(``Sampler._draw_noise``): on the GPU ``az_randn_slice_f32`` evaluates exactly the rank's elements of
the draw ``torch.randn`` would make for the whole batch (same Philox subsequences, calls and
components: bit-identical, 1 / world of the work, no full-batch tensor) and advances the generator
as the full draw would; host tensors draw the full batch and slice.  So ``world`` GPUs reproduce the
single-device result sample for sample (``init_sharded`` does the same for ``x_T``).
"""

from __future__ import annotations

import time
from collections.abc import Sequence

import torch
import torch.distributed as dist
from torch import Tensor

from .sample import Sampler

__all__ = ["shard_range", "init_sharded", "sample_sharded"]


def _world(group=None) -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(batch: int, rank: int, world: int) -> range:
    r"""Contiguous shard of ``batch`` samples owned by ``rank`` (``batch`` must divide evenly)."""
    if batch % world:
        raise ValueError(f"global batch {batch} is not divisible by the world size {world}")
    per = batch // world
    return range(rank * per, (rank + 1) * per)


@torch.no_grad()
def init_sharded(sampler: Sampler, shape: Sequence[int], *, group=None, **kwargs) -> Tensor:
    r"""This rank's slice of ``sampler.init(shape, **kwargs)``, matching the single-device draw sample for sample.  With the
    reference's default ``mean`` / ``var`` (scalars, or tensors that broadcast over the batch axis) only the rank's own rows are
    formed: ``Sampler.init`` is evaluated on the local shape with the noise drawn through ``Sampler._draw_noise`` -- on the GPU
    the rank's elements of the full-batch Philox draw (``az_randn_slice_f32``; 1 / world of the work, no full-batch ``x_T``: 201 MB
    for configs[3]), on the host the full draw sliced.  Per-sample ``mean`` / ``var`` tensors take the draw-everything-and-slice path."""
    rank, world = _world(group)
    r = shard_range(shape[0], rank, world)
    mean, var = kwargs.get("mean", 0.0), kwargs.get("var", 1.0)
    per_sample = any(torch.is_tensor(v) and v.ndim == len(shape) and v.shape[0] == shape[0] and shape[0] > 1 for v in (mean, var))
    if world == 1 or per_sample:
        full = sampler.init(shape, **kwargs)
        return full[r.start : r.stop].contiguous()
    to_kw = {k: v for k, v in kwargs.items() if k not in ("mean", "var")}
    t_T = sampler.timesteps[0]
    alpha_T, sigma_T = sampler.denoiser.schedule(t_T)
    alpha_T, sigma_T = alpha_T.to(**to_kw), sigma_T.to(**to_kw)
    if torch.is_tensor(mean):
        mean = mean.to(**to_kw)
    if torch.is_tensor(var):
        var = var.to(**to_kw)
    local = (len(r), *shape[1:])
    mean_T, std_T = alpha_T * mean, torch.sqrt(alpha_T**2 * var + sigma_T**2)  # (reference azula/sample.py:121-128, same op order)
    mean_T, std_T = mean_T.expand(local), std_T.expand(local)
    prev = sampler.shard
    sampler.shard = (rank, world)
    try:
        eps = sampler._draw_noise(torch.empty(local, dtype=mean_T.dtype, device=mean_T.device))
    finally:
        sampler.shard = prev
    return mean_T + std_T * eps


@torch.no_grad()
def sample_sharded(sampler: Sampler, x_local: Tensor, *, group=None, gather: bool = True, timings: dict | None = None,
                   **kwargs) -> Tensor:
    r"""Runs ``sampler`` on this rank's shard and all-gathers ``x0``.

    Arguments:
        x_local: This rank's slice of ``x_T`` (see :func:`init_sharded`).
        gather: If False, return the local ``x0`` only.
        timings: If a dict, the two phases are fenced with device synchronisations and their wall times are
            ACCUMULATED into ``timings["sample_ms"]`` / ``timings["allgather_ms"]`` (``bench.py`` reports them per rank).
        kwargs: Passed to the sampler (per-sample kwargs such as labels must already be local).

    Returns:
        ``x0`` of the full batch, identical on every rank (or the local shard).
    """
    rank, world = _world(group)

    def lap(key: str | None, t0: float) -> float:
        if timings is None:
            return 0.0
        if x_local.is_cuda:
            torch.cuda.synchronize(x_local.device)
        t1 = time.perf_counter()
        if key is not None:
            timings[key] = timings.get(key, 0.0) + (t1 - t0) * 1e3
        return t1

    t0 = lap(None, 0.0)
    prev = sampler.shard
    sampler.shard = (rank, world) if world > 1 else None
    try:
        x0 = sampler(x_local, **kwargs)
    finally:
        sampler.shard = prev
    t0 = lap("sample_ms", t0)
    if not gather or not (dist.is_available() and dist.is_initialized()):
        return x0
    # the collective runs whenever a process group exists -- also for a world of ONE rank, so that a single GPU
    # exercises the RCCL code path (tests/test_dist_gpu.py)
    x0 = x0.contiguous()
    out = torch.empty((world * x0.shape[0], *x0.shape[1:]), dtype=x0.dtype, device=x0.device)
    if x0.is_cuda and dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, x0, group=group)  # one RCCL all-gather over xGMI
    else:  # gloo (CPU tests, or a single-GPU rehearsal of the multi-process path)
        dist.all_gather(list(out.chunk(world)), x0, group=group)
    lap("allgather_ms", t0)
    return out
