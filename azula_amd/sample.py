r"""Reverse-diffusion samplers -- drop-in for ``azula.sample`` (DDPM / DDIM).

Same constructors, ``timesteps``, ``init``, ``__call__`` and overridable ``step`` as the reference
(``azula/sample.py:54-261``).  What changes is *how* a step executes on an AMD GPU:

reference (per step)                         | azula_amd (per step)
---------------------------------------------|---------------------------------------------------
~55 0-d scalar kernels for alpha/sigma/tau   | host table (torch-CPU, reference op order), 1 H2D
c_in*x, c_skip*x, c_out*F, +, 9 more passes  | ONE fused pass ``az_transition_f32`` (12-16 B/elem)
~1000 ATen launches for the backbone         | one hipGraph replay of hand-written kernels

The fused path is taken when the denoiser exposes a compiled program (``Denoiser._az_fused``)
and ``step`` is not overridden; any other denoiser / subclass runs the generic loop, whose
``step`` still uses the HIP transition kernel for device tensors.  Host tensors follow the
reference's torch op sequence (its CPU-runnable README configuration).
"""

from __future__ import annotations

import abc
import ctypes as C
import math
import os
from collections.abc import Iterable, Sequence
from dataclasses import dataclass, field
from typing import Callable

import torch
from torch import Tensor

from . import _lib
from ._lib import COEF_FIELDS, COEF_WORDS
from .denoise import Denoiser, axpby_wide, is_wide, require_f32_cuda
from .engine import StepGraph, Tape, transition_args

SLICED_NOISE = os.environ.get("AZ_SLICED_NOISE", "1") != "0"  # sharded batches: draw only the rank's slice of the full-batch noise
WIDE_FUSED = os.environ.get("AZ_WIDE_FUSED", "1") != "0"  # captured loop for Sampler(dtype=float64) ("0": the generic fp64 loop)

__all__ = [
    "Sampler", "DDPMSampler", "DDIMSampler", "EulerSampler", "HeunSampler", "ItoSampler", "zABSampler", "vABSampler",
    "zEABSampler", "xEABSampler", "REABSampler", "PCSampler", "FusedDenoiser", "BackboneProgram",
]


# ------------------------------------------------------------------------------- fused protocol
@dataclass
class BackboneProgram:
    r"""One compiled backbone evaluation inside the step graph.

    ``x_in`` is the NHWC (channel stride ``x_in_cs``) or flat buffer the transition kernel
    pre-scales for the next step; ``out`` receives F.  ``tape`` holds the kernels.
    """

    tape: Tape
    x_in: Tensor
    x_in_cs: int  # > 0: NHWC with this channel stride; 0: same layout as x
    out: Tensor
    f_channels: int
    f_nhwc: bool
    prepare: Callable[[dict], None] | None = None  # runs before each sampling call (e.g. upload labels)


@dataclass
class FusedDenoiser:
    r"""What a denoiser contributes to the fused step: host coefficient function + programs.

    ``programs`` has one entry (plain denoiser) or two sharing ``x_in`` (CFG: positive, negative).
    """

    coefficients: Callable[[Tensor, Tensor], dict]
    programs: list[BackboneProgram]
    guidance: float = 0.0
    clip: tuple[float, float] = (-math.inf, math.inf)
    extra: dict = field(default_factory=dict)


class Sampler(abc.ABC):
    r"""Abstract reverse diffusion sampler (reference ``azula/sample.py:54-183``)."""

    denoiser: Denoiser

    def __init__(
        self,
        start: float = 1.0,
        stop: float = 0.0,
        steps: int = 64,
        silent: bool = False,
        dtype: torch.dtype | None = None,
        device: torch.device | None = None,
    ) -> None:
        self.start = start
        self.stop = stop
        self.steps = steps
        self.silent = silent
        self.dtype = dtype
        self.device = device
        self.rng_parity = True  # draw one randn per step like the reference even when it is unused
        self.shard: tuple[int, int] | None = None  # (rank, world): see azula_amd.parallel
        self._fused_cache: dict = {}

    @property
    def timesteps(self) -> Tensor:
        return torch.linspace(self.start, self.stop, self.steps + 1, dtype=self.dtype, device=self.device)

    def invalidate(self) -> None:
        r"""Forget the captured plan and every cached hyper-parameter value: the next call re-reads weights, schedule,
        guidance / eta / temperature tensors and rebuilds the coefficient table, as after construction.  Needed only after
        writes the plan key cannot see -- ``tensor.data.copy_()``, custom kernels or other raw writes through ``data_ptr()``,
        which do not bump a tensor's version counter (ordinary in-place ops, ``load_state_dict``, ``.to()`` are tracked)."""
        self._fused_cache = {}
        _SMALL_VALUES.clear()
        # the backbones keep their own plan caches -- packed (direct / bf16x3 / Winograd-domain) copies of the weights keyed on
        # (shape, version, address), none of which a raw write changes: dropped too, so that the rebuilt loop repacks
        den = getattr(self, "denoiser", None)
        if isinstance(den, torch.nn.Module):
            for m in den.modules():
                plans = getattr(m, "_plans", None)
                if isinstance(plans, dict):
                    plans.clear()

    @torch.no_grad()
    def init(self, shape: Sequence[int], mean: float | Tensor = 0.0, var: float | Tensor = 1.0, **kwargs) -> Tensor:
        r"""x_T ~ N(alpha_T E[X], alpha_T^2 V[X] + sigma_T^2 I) (reference ``azula/sample.py:96-128``).
        ``kwargs`` go to ``Tensor.to`` (e.g. ``device="cuda"``); the draw uses the target
        device's default generator, as in the reference."""
        t_T = self.timesteps[0]
        alpha_T, sigma_T = self.denoiser.schedule(t_T)
        alpha_T, sigma_T = alpha_T.to(**kwargs), sigma_T.to(**kwargs)
        mean_T, std_T = alpha_T * mean, torch.sqrt(alpha_T**2 * var + sigma_T**2)
        mean_T, std_T = mean_T.expand(shape), std_T.expand(shape)
        return mean_T + std_T * torch.randn_like(mean_T)

    def progress_bar(self, it: Iterable) -> Iterable:
        if torch.is_tensor(it):
            it = it.unbind()
        if self.silent:
            return it
        from tqdm import tqdm

        return tqdm(it, miniters=1, unit="step", ncols=79, ascii=True)

    @torch.no_grad()
    @_lib.on_device
    def __call__(self, x: Tensor, **kwargs) -> Tensor:
        r"""Simulates the reverse process from t_T to t_0 (reference ``azula/sample.py:139-161``)."""
        if x.is_cuda and self._fusable(x):
            out = self._call_fused(x, kwargs)
            if out is not None:
                return out
        time_pairs = self.timesteps.unfold(0, 2, 1).to(device=x.device)
        x_t = x
        for t, s in self.progress_bar(time_pairs):
            x_t = self.step(x_t, t, s, **kwargs)
        return x_t

    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        raise NotImplementedError()

    def _draw_noise(self, like: Tensor, out: Tensor | None = None) -> Tensor:
        r"""One ``randn_like`` per step, as the reference (``azula/sample.py:214,259``).  When the batch
        is sharded (``self.shard = (rank, world)``) every rank draws the FULL batch from its
        identically seeded generator and keeps its slice, so an N-GPU run reproduces the
        single-device random stream sample for sample.  fp32 device tensors: ``az_randn_slice_f32`` produces exactly this
        rank's elements of that draw (1 / world of the generator work and no full-batch tensor) and advances the generator's
        Philox offset as the full draw would; ``AZ_SLICED_NOISE=0``: the full draw."""
        if self.shard is None:
            return torch.randn_like(like) if out is None else out.normal_()
        rank, world = self.shard
        if (like.is_cuda and like.dtype == torch.float32 and SLICED_NOISE and world * like.numel() < 2**31
                and _randn_slice_verified(like.device)):
            return _randn_slice(like, rank, world, out)  # this rank's elements of the full-batch draw only (bit-identical)
        full = torch.randn((world * like.shape[0], *like.shape[1:]), dtype=like.dtype, device=like.device)
        mine = full[rank * like.shape[0] : (rank + 1) * like.shape[0]]
        return mine.contiguous() if out is None else out.copy_(mine)

    # ---------------------------------------------------------------------------- shared DDPM/DDIM
    def _tau(self, alpha_t, sigma_t, alpha_s, sigma_s) -> Tensor:
        raise NotImplementedError()

    def _kernel_scalars(self, alpha_t, sigma_t, alpha_s, sigma_s):
        r"""(a_t, a_s, k_x, k_eps) of the kernel form x_s = a_s m + k_x (x_t - a_t m) + k_eps eps.
        DDPM / DDIM use it verbatim (reference op order, ``azula/sample.py:249-259``)."""
        tau = self._tau(alpha_t, sigma_t, alpha_s, sigma_s)
        return alpha_t, alpha_s, sigma_s * torch.sqrt(1 - tau) / sigma_t, sigma_s * torch.sqrt(tau)

    def _transition_scalars(self, t: Tensor, s: Tensor):
        r"""0-d scalars of one transition: the schedule at (t, s) and the kernel coefficients."""
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        return (alpha_t, sigma_t, alpha_s, sigma_s, *self._kernel_scalars(alpha_t, sigma_t, alpha_s, sigma_s))

    def _host_step(self, x_t: Tensor, mean: Tensor, t: Tensor, s: Tensor) -> Tensor:
        r"""Host tensors: the reference's op sequence (``azula/sample.py:210-214, 257-259``)."""
        _, _, _, _, a_t, a_s, k_x, k_eps = self._transition_scalars(t, s)
        x_s = a_s * mean
        x_s = x_s + k_x * (x_t - a_t * mean)
        x_s = x_s + k_eps * self._draw_noise(x_t)
        return x_s

    def _step_impl(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        q_t = self.denoiser(x_t, t, **kwargs)
        if not x_t.is_cuda:
            return self._host_step(x_t, q_t.mean, t, s)
        return self._device_transition(x_t, q_t.mean, t, s)

    def _device_transition(self, x_t: Tensor, mean: Tensor, t: Tensor, s: Tensor) -> Tensor:
        _, _, _, _, alpha_t, alpha_s, k_x, k_eps = self._transition_scalars(t, s)
        return self._device_kernel(x_t, mean, alpha_t, alpha_s, k_x, k_eps, True)

    def _device_kernel(self, x_t: Tensor, mean: Tensor, alpha_t, alpha_s, k_x, k_eps, draw: bool) -> Tensor:
        r"""x_s = alpha_s m + k_x (x_t - alpha_t m) + k_eps eps through ``az_transition_f32``; ``draw=False``
        skips the noise draw and the eps stream (the kernel's EPS=false instantiation)."""
        if is_wide(x_t, mean):
            return self._device_kernel_wide(x_t, mean, alpha_t, alpha_s, k_x, k_eps, draw)
        require_f32_cuda(x_t, type(self).__name__)
        dev = x_t.device
        zero = torch.zeros((), device=dev)
        row = torch.zeros(COEF_WORDS, dtype=torch.float32, device=dev)
        vals = {
            "c_skip": zero, "c_out": zero + 1, "alpha_t": alpha_t, "alpha_s": alpha_s, "k_x": k_x, "k_eps": k_eps,
            "clip_lo": zero - math.inf, "clip_hi": zero + math.inf,
        }
        for name, v in vals.items():
            row[COEF_FIELDS.index(name)] = v.to(device=dev, dtype=torch.float32)
        x_c = x_t.contiguous()
        mean = mean.to(x_c).contiguous()
        eps = self._draw_noise(x_c) if draw else None
        x_s = torch.empty_like(x_c)
        a = transition_args(
            x_t=x_c.data_ptr(), F=mean.data_ptr(), eps=eps.data_ptr() if draw else 0, x_s=x_s.data_ptr(), batch=1, channels=1,
            inner=x_c.numel(), f_channels=1, coef=row.data_ptr(),
        )
        _lib.call("az_transition_f32", C.byref(a), _lib.stream_ptr())
        return x_s

    def _device_kernel_wide(self, x_t: Tensor, mean: Tensor, alpha_t, alpha_s, k_x, k_eps, draw: bool) -> Tensor:
        r"""The same update in fp64 (``az_transition_f64``): fp64 latents, or an fp64 posterior mean -- what a
        ``Sampler(dtype=float64)`` produces from its first step on (the reference's promotion, ``azula/sample.py:257-259``).
        The noise is drawn in ``x_t``'s own dtype, like the reference's ``randn_like(x_t)`` (``_noise_like_update``: in the
        updated state's, fp64)."""
        dev = x_t.device
        row = torch.zeros(12, dtype=torch.float64, device=dev)
        names = ["c_in", "c_skip", "c_out", "c_time", "alpha_t", "alpha_s", "k_x", "k_eps", "c_in_next", "clip_lo", "clip_hi", "guidance"]
        vals = {"c_out": 1.0, "alpha_t": alpha_t, "alpha_s": alpha_s, "k_x": k_x, "k_eps": k_eps, "clip_lo": -math.inf, "clip_hi": math.inf}
        for name, v in vals.items():
            row[names.index(name)] = v.to(device=dev, dtype=torch.float64) if torch.is_tensor(v) else v
        x64, m64 = x_t.to(torch.float64).contiguous(), mean.to(torch.float64).contiguous()
        # (the Ito step draws randn_like(x_s), which the fp64 scalars have promoted already: azula/sample.py:427-429)
        eps = self._draw_noise(x64 if self._noise_like_update else x_t.contiguous()).to(torch.float64) if draw else None
        x_s = torch.empty_like(x64)
        a = transition_args(x_t=x64.data_ptr(), F=m64.data_ptr(), eps=eps.data_ptr() if draw else 0, x_s=x_s.data_ptr(), batch=1,
                            channels=1, inner=x64.numel(), f_channels=1, coef=row.data_ptr())
        _lib.call("az_transition_f64", C.byref(a), _lib.stream_ptr())
        return x_s

    # ---------------------------------------------------------------------------- fused path
    def _fusable(self, x: Tensor) -> bool:
        if type(self).step not in _FUSED_STEPS:
            return False  # a user subclass overrides step (guidance samplers ...): generic loop
        if x.ndim < 2:
            return False
        if x.dtype == torch.float32 and self.dtype in (None, torch.float32):
            return True
        # fp64 time grid (fp32 or fp64 latents): the captured loop with fp64 elementwise kernels (_FusedLoopWide) for the
        # samplers whose step is one evaluation + one transition
        return self._wide_fusable(x)

    # True: the step's randn_like takes the UPDATED state (already fp64 under an fp64 clock) as its model, not x_t (ItoSampler)
    _noise_like_update = False

    def _wide_fusable(self, x: Tensor) -> bool:
        r"""The one-evaluation samplers (DDPM, DDIM, Euler, Ito), the two whose tapes are built from the loop's own blocks
        (Heun, PC) and the multistep family (its fp64 weights in words 36 .. 47 of the row)."""
        tapes = (Sampler._fused_step_tapes, HeunSampler._fused_step_tapes, PCSampler._fused_step_tapes, _MultistepSampler._fused_step_tapes)
        return (self.dtype == torch.float64 and x.dtype in (torch.float32, torch.float64) and WIDE_FUSED
                and type(self)._fused_step_tapes in tapes)

    def _host_table_wide(self, fused: "FusedDenoiser") -> Tensor:
        r"""(steps * evaluations per step, 24) fp64: per evaluation the denoiser's row [c_in, c_skip, c_out, c_time, ...] and the transition's row
        [.., c_skip = 0, c_out = 1, alpha_t, alpha_s, k_x, k_eps, .., clip_lo = -inf, clip_hi = inf, ..] in the field order of
        ``az_transition_f64``'s coefficient row -- the 0-d host scalars of the fp64 time grid, in the reference's op order;
        words 24 .. 35 the clamp / CFG row, words 36 .. 47 free for the sampler (the multistep family's weights)."""
        names = ["c_in", "c_skip", "c_out", "c_time", "alpha_t", "alpha_s", "k_x", "k_eps", "c_in_next", "clip_lo", "clip_hi", "guidance"]
        ts = torch.linspace(self.start, self.stop, self.steps + 1, dtype=self.dtype)
        out = []
        f64 = lambda v: v.to(torch.float64) if torch.is_tensor(v) else float(v)  # noqa: E731
        for t, s in ts.unfold(0, 2, 1).unbind():
            for spec in self._fused_rows(t, s, fused):  # one row per denoiser evaluation, like _host_table
                row = torch.zeros(_FusedLoopWide.WORDS, dtype=torch.float64)
                co = fused.coefficients(spec["alpha"], spec["sigma"])
                for n in ("c_in", "c_skip", "c_out", "c_time"):
                    if n in co:  # (ADM reads an integer time index from the fp32 row instead of c_time)
                        row[names.index(n)] = co[n].to(torch.float64)
                # (words 4, 5 of the denoiser half are free: the two spare coefficients of a row, HeunSampler's p and q)
                row[4], row[5] = f64(spec.get("pad0", 0.0)), f64(spec.get("pad1", 0.0))
                row[names.index("clip_lo")], row[names.index("clip_hi")] = -math.inf, math.inf
                b = 12
                row[b + names.index("c_out")] = 1.0
                row[b + names.index("alpha_t")], row[b + names.index("alpha_s")] = f64(spec["a_t"]), f64(spec["a_s"])
                row[b + names.index("k_x")], row[b + names.index("k_eps")] = f64(spec["k_x"]), f64(spec["k_eps"])
                row[b + names.index("clip_lo")], row[b + names.index("clip_hi")] = -math.inf, math.inf
                # words 24 .. 35: the CLAMP row of the posterior mean (a transition row that only clips: c_out = 1, alpha_s = 1)
                # with the constants -1 / +1 and the guidance of the CFG combination in its unused slots
                b = 24
                row[b + names.index("c_out")], row[b + names.index("alpha_s")] = 1.0, 1.0
                row[b + names.index("clip_lo")], row[b + names.index("clip_hi")] = fused.clip
                row[b + names.index("c_time")], row[b + names.index("c_in_next")] = -1.0, 1.0
                row[b + names.index("guidance")] = fused.guidance
                out.append(row)
        return torch.stack(out)

    # -- what a step looks like inside the captured graph ----------------------------------------------------------
    def _fused_rows(self, t: Tensor, s: Tensor, fused: "FusedDenoiser") -> list[dict]:
        r"""One dict per DENOISER EVALUATION of the step t -> s, in execution order: the 0-d host scalars of the table
        row that evaluation reads (``alpha`` / ``sigma``: where the denoiser is evaluated; ``a_t, a_s, k_x, k_eps`` of
        the kernel form x' = a_s m + k_x (x - a_t m) + k_eps eps; optional ``pad0, pad1``)."""
        alpha_t, sigma_t, _, _, ka_t, ka_s, k_x, k_eps = self._transition_scalars(t, s)
        return [dict(alpha=alpha_t, sigma=sigma_t, a_t=ka_t, a_s=ka_s, k_x=k_x, k_eps=k_eps)]

    def _noise_draws(self) -> int:
        r"""randn_like draws per step that the kernels READ (the reference's generator calls, in order)."""
        return 1 if self._needs_noise() else 0

    def _fused_structure(self) -> tuple:
        r"""Everything about the sampler that changes the SHAPE of the captured graph (not just table values)."""
        return (type(self).__name__, self.steps, self._noise_draws(), self.rng_parity)

    def _host_table(self, fused: FusedDenoiser) -> Tensor:
        r"""(steps * rows_per_step, 16) fp32 table of AzStepCoef rows, from 0-d CPU tensors in reference op order."""
        ts = torch.linspace(self.start, self.stop, self.steps + 1, dtype=self.dtype)
        col = {n: i for i, n in enumerate(COEF_FIELDS)}
        out = []
        for i, (t, s) in enumerate(ts.unfold(0, 2, 1).unbind()):
            for spec in self._fused_rows(t, s, fused):
                row = torch.zeros(COEF_WORDS, dtype=torch.float32)
                irow = row.view(torch.int32)  # integer slots (time_index, step) are bit-cast in place
                for name, v in fused.coefficients(spec["alpha"], spec["sigma"]).items():
                    if name == "time_index":
                        irow[col[name]] = int(v)
                    else:
                        row[col[name]] = v.to(torch.float32)
                row[col["alpha_t"]], row[col["alpha_s"]] = spec["a_t"], spec["a_s"]
                row[col["k_x"]], row[col["k_eps"]] = spec["k_x"], spec["k_eps"]
                row[col["clip_lo"]], row[col["clip_hi"]] = fused.clip
                row[col["guidance"]] = fused.guidance
                irow[col["step"]] = i
                row[14], row[15] = spec.get("pad0", 0.0), spec.get("pad1", 0.0)
                out.append(row)
        rows = torch.stack(out)
        rows[:-1, col["c_in_next"]] = rows[1:, col["c_in"]]  # each evaluation pre-scales the NEXT evaluation's input
        return rows

    def _fused_step_tapes(self, loop: "_FusedLoop") -> list[Tape]:
        r"""The kernels of ONE step (default: one denoiser evaluation + the fused transition, in place on ``loop.x``).
        Samplers whose steps differ in their buffer addresses return one tape per phase of that cycle."""
        tape = Tape()
        loop.add_evaluation(tape)
        loop.add_transition(tape, x_t=loop.x, x_s=loop.x, eps=loop.noise[0] if loop.noise else None)
        return [tape]

    def _needs_noise(self) -> bool:
        raise NotImplementedError()

    def _draws_unused_noise(self) -> bool:
        r"""True if the reference's step draws a randn_like that this configuration never reads (DDIM with eta = 0)."""
        return False

    def _fused_upload_extra(self, loop: "_FusedLoop") -> None:
        r"""Sampler-specific device tables, refreshed together with the coefficient table."""

    def _fused_reset(self, loop: "_FusedLoop") -> None:
        r"""Per-call state of the loop's extra buffers (history rings ...)."""

    def _hyper(self) -> tuple:
        r"""Every scalar hyper-parameter of the sampler (start, stop, steps, eta, temperature, order, ...): whatever
        feeds ``_kernel_scalars`` / ``_host_table`` is a plain attribute, so a change of any of them is seen."""
        return _attr_key(vars(self))

    def _call_fused(self, x: Tensor, kwargs: dict) -> Tensor | None:
        r"""The captured loop is cached per sampler, keyed on everything that is BAKED into it: the latent's shape and
        device, the structure of the keyword arguments, the step count / noise use, and a fingerprint of the denoiser
        (address, version counter and dtype of every parameter and buffer, train/eval flags).  The reference re-reads
        weights, guidance and hyper-parameters on every call (``azula/sample.py:139-161``); so does this: after
        ``load_state_dict`` / ``.half()`` / ``.train()`` the plan is rebuilt, and values that only live in the
        coefficient table (guidance, eta, temperature, start / stop, the schedule) are re-uploaded by ``run``."""
        dev = x.device
        g = kwargs.get("guidance")
        if torch.is_tensor(g) and g.numel() != 1:
            return None  # per-sample guidance: generic loop
        wide = not (x.dtype == torch.float32 and self.dtype in (None, torch.float32))
        key = (
            tuple(x.shape), str(dev), _kwargs_signature(kwargs), self._fused_structure(), id(self.denoiser),
            module_fingerprint(self.denoiser), str(x.dtype), wide,
        )
        ent = self._fused_cache.get(key)
        if ent is None:
            self._fused_cache = {}  # drop the stale plan first: one live plan per sampler keeps HBM use bounded
            cur = torch.zeros(COEF_WORDS, dtype=torch.float32, device=dev)
            fused = self.denoiser._az_fused(torch.empty(x.shape, dtype=torch.float32, device=dev) if wide else x, kwargs, cur)
            if fused is None:
                return None
            if wide:
                p0 = fused.programs[0]
                Cc = x.shape[1] if x.ndim > 2 else 1
                if p0.f_nhwc or p0.f_channels < Cc or len(fused.programs) > 2:
                    return None  # (a backbone that hands F over channels-last: the generic fp64 loop)
                ent = _FusedLoopWide(self, fused, x, cur)
            else:
                ent = _FusedLoop(self, fused, x, cur)
            self._fused_cache = {key: ent}
        return ent.run(x, kwargs)


def module_fingerprint(module: torch.nn.Module) -> tuple:
    r"""Identity of a module's state as the compiled plans see it: (address, in-place version, dtype) of every
    parameter and buffer plus the train/eval flags.  Packed weight copies (direct, Winograd, half) are valid exactly
    as long as this tuple is unchanged."""
    tensors = tuple((t.data_ptr(), t._version, t.dtype) for t in (*module.parameters(), *module.buffers()))
    return tensors + tuple(m.training for m in module.modules())


def _schedule_key(sched) -> tuple:
    r"""Scalar attributes of a schedule object (alpha_min, sigma_min, gamma, ...) and, for ``nn.Module`` schedules,
    the fingerprint of their tensors."""
    return _attr_key(getattr(sched, "__dict__", {})) + (module_fingerprint(sched) if isinstance(sched, torch.nn.Module) else ())


def _value_key(v):
    r"""A hashable stand-in for one attribute value, or ``_SKIP``: numbers / strings / dtypes by value, small tensors by
    VALUE (eta, temperature, alpha_min held as 0-d or short tensors: the reference re-reads them on every call), larger
    tensors by (address, version), tuples / lists recursively.  Modules and other objects are not hyper-parameters."""
    if isinstance(v, (int, float, bool, str, type(None), torch.dtype)):
        return v
    if torch.is_tensor(v):
        if v.numel() <= 16:
            # `.tolist()` on a device tensor is a blocking sync: read the value only when the tensor's identity
            # (object, storage address, version counter) changes.  Writes through `.data` do not bump the version counter and
            # are NOT tracked -- neither here nor by the (address, version) key of larger tensors (documented in DESIGN.md).
            ident = (v.data_ptr(), v._version, tuple(v.shape), v.dtype, v.device)
            hit = _SMALL_VALUES.get(id(v))
            if hit is None or hit[0] != ident or hit[1]() is not v:
                import weakref

                if len(_SMALL_VALUES) > 256:
                    for k in [k for k, h in _SMALL_VALUES.items() if h[1]() is None]:
                        del _SMALL_VALUES[k]
                hit = (ident, weakref.ref(v), tuple(v.detach().reshape(-1).tolist()))
                _SMALL_VALUES[id(v)] = hit
            return ("t", tuple(v.shape), str(v.dtype), hit[2])
        return ("T", v.data_ptr(), v._version, tuple(v.shape), str(v.dtype))
    if isinstance(v, (tuple, list)):
        items = tuple(_value_key(x) for x in v)
        return _SKIP if any(x is _SKIP for x in items) else ("seq", items)
    return _SKIP


_SKIP = object()
_SMALL_VALUES: dict = {}  # id(tensor) -> ((address, version, shape, dtype, device), weakref, value tuple)


def _attr_key(attrs: dict) -> tuple:
    out = []
    for k, v in sorted(attrs.items()):
        if k.startswith("_fused") or isinstance(v, torch.nn.Module):
            continue
        kv = _value_key(v)
        if kv is not _SKIP:
            out.append((k, kv))
    return tuple(out)


def _kwargs_signature(kw) -> tuple:
    r"""Structure of the keyword arguments (names, nesting, which values are None) -- not their values: label and
    guidance VALUES are uploaded per call, their presence decides which programs exist."""
    if isinstance(kw, dict):
        return tuple((k, _kwargs_signature(v)) for k, v in sorted(kw.items()))
    if kw is None:
        return ("none",)
    if torch.is_tensor(kw):
        return ("tensor", tuple(kw.shape))
    return ("value",)


_SLICE_VERIFIED: dict = {}  # (device index, torch version) -> bool


def _randn_slice_verified(dev: torch.device) -> bool:
    r"""One-time self-check per (device, torch version): :func:`_randn_slice` re-derives ATen's launch policy (grid rule, unroll 4,
    offset arithmetic) from the torch it was written against.  A torch whose policy differs would still hand out valid noise,
    but the 1-GPU == N-GPU bit-reproducibility of ``parallel.py`` would break silently.  So the first sharded draw on a device
    compares two slices -- a tensor under one grid's worth of elements and one that spans several grid iterations -- and the
    generator's offset afterwards against ``torch.randn`` of the full shape, under a saved / restored generator state; on any
    mismatch it warns once and every later sharded draw takes the draw-and-slice path."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.__version__)
    ok = _SLICE_VERIFIED.get(key)
    if ok is None:
        gen = torch.cuda.default_generators[idx]
        saved = gen.get_state()
        ok = True
        try:
            for total, world, rank in ((3 * 1000, 3, 1), (3 * (1 << 20) + 3 * 4096, 3, 2), (2 * 65536, 2, 0)):
                gen.manual_seed(1234567)
                gen.set_offset(8)
                full = torch.randn(total, device=dev)
                off_full = gen.get_offset()
                gen.manual_seed(1234567)
                gen.set_offset(8)
                n = total // world
                mine = _randn_slice(torch.empty(n, device=dev), rank, world)
                ok = ok and gen.get_offset() == off_full and bool(torch.equal(mine, full[rank * n : (rank + 1) * n]))
        except Exception:  # noqa: BLE001  (a generator API that moved: same verdict)
            ok = False
        finally:
            gen.set_state(saved)
        _SLICE_VERIFIED[key] = ok
        if not ok:
            import warnings

            warnings.warn(
                f"azula_amd: the sliced Philox draw does not reproduce torch.randn on torch {torch.__version__} (device {idx}); "
                "sharded sampling falls back to drawing the full batch on every rank and slicing (same result, more work)",
                RuntimeWarning, stacklevel=3)
    return ok


def _randn_slice(like: Tensor, rank: int, world: int, out: Tensor | None = None) -> Tensor:
    r"""Elements [rank n, (rank + 1) n) of ``torch.randn(world * n)`` on ``like``'s device, without drawing the rest: ATen's
    launch policy for the FULL tensor (``ATen/native/cuda/DistributionTemplates.h: calc_execution_policy``) fixes which Philox
    subsequence / call / component produces every element; the kernel evaluates exactly those."""
    dev = like.device
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    n = like.numel()
    total = world * n
    props = torch.cuda.get_device_properties(dev)
    grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (total + 255) // 256)
    threads = 256 * grid
    seed, offset = gen.initial_seed(), gen.get_offset()
    dst = out if (out is not None and out.is_contiguous()) else torch.empty_like(like, memory_format=torch.contiguous_format)
    with torch.cuda.device(dev):
        _lib.call("az_randn_slice_f32", dst.data_ptr(), seed, offset, threads, rank * n, n, _lib.stream_ptr())
    gen.set_offset(offset + ((total - 1) // (threads * 4) + 1) * 4)  # (the full draw's counter_offset)
    if out is not None and dst is not out:
        out.copy_(dst)
        return out
    return dst


class _FusedLoop:
    r"""Static buffers + step tapes + hipGraphs for one (sampler, denoiser, shape).

    A STEP is ``rows_per_step`` denoiser evaluations, each ``az_step_begin`` (next table row -> ``cur``) + the backbone
    programs + an elementwise update; the sampler lays its step out on a tape (``Sampler._fused_step_tapes``) and the
    loop replays the captured graph of ``period`` consecutive steps (1 except for the multistep family, whose history
    ring makes the buffer addresses cycle with period ``order``)."""

    @property
    def sampler(self) -> "Sampler":
        return self._sampler()

    def __init__(self, sampler: Sampler, fused: FusedDenoiser, x: Tensor, cur: Tensor) -> None:
        # (a weak reference: the loop lives in sampler._fused_cache -- a strong one would make a cycle, and a dropped plan's pool,
        #  tens of GB of HBM at ADM sizes, would wait for the cycle collector instead of going with the last reference)
        import weakref

        self._sampler, self.fused, self.cur = weakref.ref(sampler), fused, cur
        dev = x.device
        self.x = torch.empty_like(x, memory_format=torch.contiguous_format)
        # noise[k]: the k-th randn_like of a step that a kernel reads.  With DDIM eta = 0 nothing reads it, but the
        # reference still draws one per step: `rng_parity` keeps that draw (into `dummy`) so that the generator state
        # after sampling matches, while the transition moves 12 B/element instead of 16.
        self.noise = [torch.empty_like(self.x) for _ in range(sampler._noise_draws())]
        self.dummy = torch.empty_like(self.x) if (not self.noise and sampler.rng_parity and sampler._draws_unused_noise()) else None
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.table_key = None
        B = x.shape[0]
        Cc = x.shape[1] if x.ndim > 2 else 1
        self.B, self.C, self.inner = B, Cc, x.numel() // (B * Cc)
        self.n_rows = len(sampler._host_table(fused))
        self.table = torch.zeros(self.n_rows, COEF_WORDS, dtype=torch.float32, device=dev)
        self.keep: list = []
        self.step_tapes = sampler._fused_step_tapes(self)
        self.period = len(self.step_tapes)
        self.graphs: dict[int, StepGraph] = {}

    # -- tape building blocks (used by Sampler._fused_step_tapes) -----------------------------------------------
    def add_evaluation(self, tape: Tape) -> None:
        r"""az_step_begin + every backbone program: F (and F_neg) of the NEXT table row's evaluation."""
        tape.add("az_step_begin", self.cur.data_ptr(), self.table.data_ptr(), self.counter.data_ptr(), self.n_rows)
        for p in self.fused.programs:
            tape.extend(p.tape)

    def add_transition(self, tape: Tape, *, x_t: Tensor, x_s: Tensor, eps: Tensor | None = None, mean_out: Tensor | None = None,
                       write_xin: bool = True) -> None:
        r"""x_s = a_s m + k_x (x_t - a_t m) + k_eps eps with m = clip(c_skip x_t + c_out F) [CFG-combined], and the next
        evaluation's pre-scaled backbone input c_in' x_s in the backbone's layout -- one pass (``az_transition_f32``)."""
        p0 = self.fused.programs[0]
        a = transition_args(
            x_t=x_t.data_ptr(), F=p0.out.data_ptr(),
            F_neg=self.fused.programs[1].out.data_ptr() if len(self.fused.programs) > 1 else None,
            eps=eps.data_ptr() if eps is not None else None, x_s=x_s.data_ptr(),
            mean_out=mean_out.data_ptr() if mean_out is not None else None,
            xin_next=p0.x_in.data_ptr() if write_xin else None, batch=self.B, channels=self.C, inner=self.inner,
            f_channels=p0.f_channels, f_nhwc=int(p0.f_nhwc), nhwc_pad=p0.x_in_cs if write_xin else 0, coef=self.cur.data_ptr(),
        )
        tape.add("az_transition_f32", C.byref(a), keep=[a])

    def add_input_relayout(self, tape: Tape | None, x: Tensor, scale_ptr: int) -> None:
        r"""backbone input <- scale * x in the backbone's layout (NHWC with a padded channel stride, or flat)."""
        p0 = self.fused.programs[0]
        if p0.x_in_cs > 0:
            args = ("az_nchw_to_nhwc_f32", p0.x_in.data_ptr(), x.data_ptr(), scale_ptr, self.B, self.C, self.inner, p0.x_in_cs)
        else:
            args = ("az_scale_f32", p0.x_in.data_ptr(), x.data_ptr(), scale_ptr, x.numel())
        if tape is None:
            _lib.call(*args, _lib.stream_ptr())
        else:
            tape.add(*args)

    def add_spare_axpby(self, tape: Tape, y: Tensor, x: Tensor, z: Tensor) -> None:
        r"""y = p x + q z with the current row's two spare coefficients (``pad0``, ``pad1`` of ``Sampler._fused_rows``)."""
        tape.add("az_axpby_f32", y.data_ptr(), self.coef_ptr(14), x.data_ptr(), self.coef_ptr(15), z.data_ptr(), 1, x.numel(), 0)

    def coef_ptr(self, name_or_word) -> int:
        r"""Device address of one float of the current row (``cur``): kernels that take coefficient POINTERS read the
        step's value through it, so the captured graph needs no per-step update."""
        w = COEF_FIELDS.index(name_or_word) if isinstance(name_or_word, str) else int(name_or_word)
        return self.cur.data_ptr() + 4 * w

    # -- compatibility views (bench.py / tools / tests) -------------------------------------------------------------
    @property
    def tape(self) -> Tape:
        return self.step_tapes[0]

    @property
    def graph(self) -> StepGraph | None:
        return self.graphs.get(self.period)

    @property
    def eps(self) -> Tensor | None:
        return self.noise[0] if self.noise else self.dummy

    def _upload_table(self, kwargs: dict) -> None:
        s = self.sampler
        if len(self.fused.programs) > 1:  # CFG: the guidance VALUE is re-read on every call (reference cfg.py:63-65)
            self.fused.guidance = float(kwargs.get("guidance", 1.0))
        sched = s.denoiser.schedule
        key = (s._hyper(), id(sched), _schedule_key(sched), self.fused.guidance, self.fused.clip)
        if key != self.table_key:
            self.table.copy_(s._host_table(self.fused))
            s._fused_upload_extra(self)
            self.table_key = key

    def run(self, x: Tensor, kwargs: dict) -> Tensor:
        s = self.sampler
        self._upload_table(kwargs)
        for p in self.fused.programs:
            if p.prepare is not None:
                p.prepare(kwargs)
        self.x.copy_(x)
        self.counter.zero_()
        s._fused_reset(self)
        stream = _lib.stream_ptr()
        # backbone input of the first evaluation: c_in[0] * x_T, in the backbone's layout
        self.add_input_relayout(None, self.x, self.table[0, COEF_FIELDS.index("c_in")].data_ptr())
        for g in s.progress_bar(range(0, s.steps, self.period)):
            n = min(self.period, s.steps - g)
            for buf in self.noise:
                s._draw_noise(buf, out=buf)  # same generator calls, in the same order, as the reference's randn_like
            if self.dummy is not None:
                s._draw_noise(self.dummy, out=self.dummy)
            graph = self.graphs.get(n)
            if graph is None:  # first group of this length: run eagerly (loads code objects), then capture for the next
                for t in self.step_tapes[:n]:
                    t.run(stream)
                cat = Tape()
                for t in self.step_tapes[:n]:
                    cat.extend(t)
                self.graphs[n] = StepGraph(cat, x.device)
            else:
                graph.launch()
        return self.x.clone()


class _FusedLoopWide(_FusedLoop):
    r"""The captured loop for ``Sampler(dtype=torch.float64)`` (reference ``azula/sample.py:69-94``, ``azula/denoise.py:306-322``):
    the schedule scalars are fp64 tensors of shape (1, ..., 1), so the latents and every elementwise statement of a step are
    fp64 while the backbone keeps fp32.  Per step: ``az_step_begin`` (fp32 row: the backbone's time embedding) +
    ``az_step_row_f64`` (fp64 rows) -> ``az_scale_f64_to_f32`` (backbone input) -> backbone tape -> ``az_axpby_f64`` (posterior
    mean) -> ``az_transition_f64`` -- the kernels of the generic fp64 loop, reading their coefficients through pointers into
    the device-resident current row, as ONE hipGraph per step.  Noise is drawn in the dtype of x_t like the reference's
    ``randn_like(x_t)``: fp32 in the first step of an fp32 input (x_t is promoted by that step), fp64 afterwards."""

    WORDS = 48  # [0, 12) the denoiser's row, [12, 24) the transition's, [24, 36) clamp / CFG constants, [36, 48) the sampler's own

    def __init__(self, sampler: "Sampler", fused: FusedDenoiser, x: Tensor, cur: Tensor) -> None:
        dev = x.device
        self.in_dtype = x.dtype
        self.cur64 = torch.zeros(self.WORDS, dtype=torch.float64, device=dev)
        n_rows = len(sampler._host_table(fused))
        self.table64 = torch.zeros(n_rows, self.WORDS, dtype=torch.float64, device=dev)
        self.mean64 = torch.empty(x.shape, dtype=torch.float64, device=dev)
        p0 = fused.programs[0]
        # (a backbone that reads its input channels-last: fp64 -> fp32 flat, then the layout pass of the fp32 loop)
        self.xin32 = torch.empty(x.shape, dtype=torch.float32, device=dev) if p0.x_in_cs > 0 else None
        # (CFG: the negative program's posterior mean and the difference of the two)
        self.cfg64 = [torch.empty(x.shape, dtype=torch.float64, device=dev) for _ in range(2)] if len(fused.programs) == 2 else None
        super().__init__(sampler, fused, torch.empty(x.shape, dtype=torch.float64, device=dev), cur)
        n32 = min(1, len(self.noise)) if x.dtype == torch.float32 else 0
        self.noise32 = [torch.empty(x.shape, dtype=torch.float32, device=dev) for _ in range(n32)]
        self.dummy32 = torch.empty(x.shape, dtype=torch.float32, device=dev) if (self.dummy is not None and x.dtype == torch.float32) else None

    def _c64(self, name: str, block: int = 0) -> int:
        names = ["c_in", "c_skip", "c_out", "c_time", "alpha_t", "alpha_s", "k_x", "k_eps", "c_in_next", "clip_lo", "clip_hi", "guidance"]
        return self.cur64.data_ptr() + 8 * (names.index(name) + 12 * int(block))

    def add_evaluation(self, tape: Tape) -> None:
        r"""The evaluation reads the state the previous transition of the tape wrote (``x`` at the head of a step): the fp32
        loop's transition kernel leaves the pre-scaled backbone input behind, here it is a pass of its own."""
        p0 = self.fused.programs[0]
        src = getattr(self, "_src", None)
        src = self.x if src is None else src
        tape.add("az_step_begin", self.cur.data_ptr(), self.table.data_ptr(), self.counter.data_ptr(), self.n_rows)
        tape.add("az_step_row_f64", self.cur64.data_ptr(), self.table64.data_ptr(), self.counter.data_ptr(), self.n_rows, self.WORDS)
        if self.xin32 is not None:
            tape.add("az_scale_f64_to_f32", self.xin32.data_ptr(), src.data_ptr(), self._c64("c_in"), 1, src.numel(), 0)
            tape.add("az_nchw_to_nhwc_f32", p0.x_in.data_ptr(), self.xin32.data_ptr(), None, self.B, self.C, self.inner, p0.x_in_cs)
        else:
            tape.add("az_scale_f64_to_f32", p0.x_in.data_ptr(), src.data_ptr(), self._c64("c_in"), 1, src.numel(), 0)
        for p in self.fused.programs:
            tape.extend(p.tape)

    def add_mean(self, tape: Tape, x_t: Tensor, mean: Tensor) -> None:
        r"""``mean`` <- the denoiser's posterior mean in fp64: c_skip x_t + c_out F (the first channels of a wider F: one launch
        per sample), clipped where the denoiser clips (``az_transition_f64`` under the clamp row, in place), and for CFG
        pos + g (pos - neg) of the two clipped means -- the statements of the per-statement fp64 loop
        (``denoise.py`` / ``plugins/adm/__init__.py`` / ``guidance/cfg.py``, ``is_wide`` branches), reading their scalars
        through pointers into the current fp64 row."""
        n = x_t.numel()
        clip = self.fused.clip != (-math.inf, math.inf)

        def raw(dst: Tensor, p: BackboneProgram) -> None:
            if p.f_channels == self.C:
                tape.add("az_axpby_f64", dst.data_ptr(), self._c64("c_skip"), x_t.data_ptr(), self._c64("c_out"), p.out.data_ptr(), 1, 1, n, 0)
            else:
                per = self.C * self.inner
                for b in range(self.B):
                    tape.add("az_axpby_f64", dst.data_ptr() + 8 * b * per, self._c64("c_skip"), x_t.data_ptr() + 8 * b * per, self._c64("c_out"),
                             p.out.data_ptr() + 4 * b * p.f_channels * self.inner, 1, 1, per, 0)
            if clip:
                a = transition_args(x_t=dst.data_ptr(), F=dst.data_ptr(), x_s=dst.data_ptr(), batch=1, channels=1, inner=n, f_channels=1,
                                    coef=self.cur64.data_ptr() + 8 * 24)
                tape.add("az_transition_f64", C.byref(a), keep=[a])

        progs = self.fused.programs
        if len(progs) == 1:
            raw(mean, progs[0])
            return
        neg, diff = self.cfg64
        raw(mean, progs[0])
        raw(neg, progs[1])
        one, minus, g = self._c64("c_in_next", 2), self._c64("c_time", 2), self._c64("guidance", 2)
        tape.add("az_axpby_f64", diff.data_ptr(), one, mean.data_ptr(), minus, neg.data_ptr(), 0, 1, n, 0)
        tape.add("az_axpby_f64", mean.data_ptr(), one, mean.data_ptr(), g, diff.data_ptr(), 0, 1, n, 0)

    def add_transition(self, tape: Tape, *, x_t: Tensor, x_s: Tensor, eps: Tensor | None = None, mean_out: Tensor | None = None,
                       write_xin: bool = True) -> None:
        n = x_t.numel()
        mean = self.mean64 if mean_out is None else mean_out  # (Heun keeps the first evaluation's posterior mean)
        self.add_mean(tape, x_t, mean)
        a = transition_args(x_t=x_t.data_ptr(), F=mean.data_ptr(), eps=eps.data_ptr() if eps is not None else None,
                            x_s=x_s.data_ptr(), batch=1, channels=1, inner=n, f_channels=1, coef=self.cur64.data_ptr() + 8 * 12)
        tape.add("az_transition_f64", C.byref(a), keep=[a])
        self._src = x_s

    def add_spare_axpby(self, tape: Tape, y: Tensor, x: Tensor, z: Tensor) -> None:
        tape.add("az_axpby_f64", y.data_ptr(), self.cur64.data_ptr() + 8 * 4, x.data_ptr(), self.cur64.data_ptr() + 8 * 5, z.data_ptr(), 0, 1,
                 x.numel(), 0)

    def _upload_table(self, kwargs: dict) -> None:
        before = self.table_key
        super()._upload_table(kwargs)
        if self.table_key != before:
            self.table64.copy_(self.sampler._host_table_wide(self.fused))

    def run(self, x: Tensor, kwargs: dict) -> Tensor:
        s = self.sampler
        self._upload_table(kwargs)
        for p in self.fused.programs:
            if p.prepare is not None:
                p.prepare(kwargs)
        self.x.copy_(x)  # (fp32 -> fp64 is exact: what the first fp64 multiplication of the reference's step does)
        self.counter.zero_()
        s._fused_reset(self)
        stream = _lib.stream_ptr()
        if self.period > 1:  # (the multistep family: `period` consecutive steps per graph, no noise)
            assert not self.noise and self.dummy is None
            for g in s.progress_bar(range(0, s.steps, self.period)):
                n = min(self.period, s.steps - g)
                graph = self.graphs.get(n)
                if graph is None:
                    cat = Tape()
                    for t in self.step_tapes[:n]:
                        t.run(stream)
                        cat.extend(t)
                    self.graphs[n] = StepGraph(cat, x.device)
                else:
                    graph.launch()
            return self.x.clone()
        for g in s.progress_bar(range(s.steps)):
            # (randn_like(x_t): fp32 only for the FIRST draw of an fp32 input -- the update it feeds promotes the state; a
            #  sampler whose draw is modelled on the updated state never draws fp32)
            first32 = g == 0 and self.in_dtype == torch.float32 and not s._noise_like_update
            for k, buf in enumerate(self.noise):
                if first32 and k == 0:
                    s._draw_noise(self.noise32[k], out=self.noise32[k])
                    buf.copy_(self.noise32[k])
                else:
                    s._draw_noise(buf, out=buf)
            if self.dummy is not None:
                d = self.dummy32 if first32 else self.dummy
                s._draw_noise(d, out=d)
            graph = self.graphs.get(1)
            if graph is None:
                self.step_tapes[0].run(stream)
                self.graphs[1] = StepGraph(self.step_tapes[0], x.device)
            else:
                graph.launch()
        return self.x.clone()


class DDPMSampler(Sampler):
    r"""DDPM sampler (reference ``azula/sample.py:186-216``):
    x_s = alpha_s mu + sigma_s sqrt(1 - tau)/sigma_t (x_t - alpha_t mu) + sigma_s sqrt(tau) eps,
    tau = 1 - (alpha_t/alpha_s * sigma_s/sigma_t)^2."""

    def __init__(self, denoiser: Denoiser, **kwargs) -> None:
        super().__init__(**kwargs)
        self.denoiser = denoiser

    def _tau(self, alpha_t, sigma_t, alpha_s, sigma_s) -> Tensor:
        return 1 - (alpha_t / alpha_s * sigma_s / sigma_t) ** 2

    def _needs_noise(self) -> bool:
        return True

    @_lib.on_device
    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        return self._step_impl(x_t, t, s, **kwargs)


class DDIMSampler(Sampler):
    r"""DDIM sampler (reference ``azula/sample.py:219-261``): as DDPM with
    tau <- clip(eta * tau, 0, 1).  ``eta = 0`` is deterministic, ``eta = 1`` equals DDPM."""

    def __init__(self, denoiser: Denoiser, eta: float = 0.0, **kwargs) -> None:
        super().__init__(**kwargs)
        self.denoiser = denoiser
        self.eta = eta

    def _tau(self, alpha_t, sigma_t, alpha_s, sigma_s) -> Tensor:
        tau = 1 - (alpha_t / alpha_s * sigma_s / sigma_t) ** 2
        return torch.clip(self.eta * tau, min=0, max=1)

    def _needs_noise(self) -> bool:
        # eta = 0 => tau = 0 => k_eps = 0: the noise term vanishes.  The reference still draws
        # randn_like (advancing the RNG); `rng_parity` keeps that draw without reading it.
        return self.eta != 0

    def _draws_unused_noise(self) -> bool:
        return self.eta == 0

    @_lib.on_device
    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        return self._step_impl(x_t, t, s, **kwargs)


# ------------------------------------------------------------------------------- next samplers
def _lin2(a: Tensor, x: Tensor, b: Tensor, y: Tensor) -> Tensor:
    r"""a * x + b * y with 0-d coefficient tensors: ``az_axpby_f32`` on device tensors."""
    if not x.is_cuda:
        return a * x + b * y
    if is_wide(x, y):
        return axpby_wide(a, x, b, y)
    require_f32_cuda(x, "sampler")
    x, y = x.contiguous(), y.to(x).contiguous()
    out = torch.empty_like(x)
    a = a.to(device=x.device, dtype=torch.float32).reshape(1)
    b = b.to(device=x.device, dtype=torch.float32).reshape(1)
    _lib.call("az_axpby_f32", out.data_ptr(), a.data_ptr(), x.data_ptr(), b.data_ptr(), y.data_ptr(), 1, x.numel(), 0, _lib.stream_ptr())
    return out


class EulerSampler(Sampler):
    r"""Explicit Euler (1st order) sampler (reference ``azula/sample.py:264-305``):
    z_t = (x_t - alpha_t mu) / sigma_t;  x_s = alpha_s/alpha_t x_t + alpha_s (sigma_s/alpha_s - sigma_t/alpha_t) z_t.
    The update is linear in (x_t, mu), so on the device it is the same fused transition kernel
    with folded coefficients (tolerance-level, not bit-level, parity with the reference)."""

    def __init__(self, denoiser: Denoiser, **kwargs) -> None:
        super().__init__(**kwargs)
        self.denoiser = denoiser

    @staticmethod
    def _euler_coefficients(alpha_t, sigma_t, alpha_s, sigma_s):
        r"""x_s = A x_t + B mu."""
        c = alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t)
        return alpha_s / alpha_t + c / sigma_t, -c * alpha_t / sigma_t

    def _kernel_scalars(self, alpha_t, sigma_t, alpha_s, sigma_s):
        A, B = self._euler_coefficients(alpha_t, sigma_t, alpha_s, sigma_s)
        zero = torch.zeros_like(A)
        return zero, B, A, zero  # x_s = B m + A (x_t - 0 m)

    def _needs_noise(self) -> bool:
        return False

    def _host_step(self, x_t, mean, t, s):
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        z_t = (x_t - alpha_t * mean) / sigma_t
        return alpha_s / alpha_t * x_t + alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t) * z_t

    @_lib.on_device
    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        return self._step_impl(x_t, t, s, **kwargs)


class HeunSampler(EulerSampler):
    r"""Explicit Heun (2nd order) sampler (reference ``azula/sample.py:308-352``): an Euler predictor,
    a second denoiser call at ``s`` and the trapezoidal corrector.  Two backbone evaluations per step.

    Fused form (one graph replay per step, two table rows): with c = alpha_s (sigma_s/alpha_s - sigma_t/alpha_t),

    * row A (time t): x_p = Euler(x_t, m_t), also keeps m_t (``mean_out``) and x_t;
    * row B (time s): e = p x_t + q m_t (``az_axpby_f32`` with p, q read from the row's spare words), then the
      transition kernel on (x_p, F_s) with eps = e, k_eps = 1:
      x_s = (alpha_s/alpha_t + c/(2 sigma_t)) x_t - c alpha_t/(2 sigma_t) m_t + c/(2 sigma_s) x_p - c alpha_s/(2 sigma_s) m_s,
      which is the reference's update with z_t, z_s expanded (tolerance-level parity, like Euler)."""

    def _fused_rows(self, t, s, fused):
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        A, B = self._euler_coefficients(alpha_t, sigma_t, alpha_s, sigma_s)
        c = alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t)
        zero, one = torch.zeros_like(c), torch.ones_like(c)
        return [
            dict(alpha=alpha_t, sigma=sigma_t, a_t=zero, a_s=B, k_x=A, k_eps=zero),
            dict(alpha=alpha_s, sigma=sigma_s, a_t=zero, a_s=-c * alpha_s / (2 * sigma_s), k_x=c / (2 * sigma_s), k_eps=one,
                 pad0=alpha_s / alpha_t + c / (2 * sigma_t), pad1=-c * alpha_t / (2 * sigma_t)),
        ]

    def _fused_step_tapes(self, loop):
        xp, m, e = (torch.empty_like(loop.x) for _ in range(3))
        loop.keep += [xp, m, e]
        tape = Tape()
        loop.add_evaluation(tape)
        loop.add_transition(tape, x_t=loop.x, x_s=xp, mean_out=m)  # x_p (+ the backbone input c_in(s) x_p), m_t
        loop.add_evaluation(tape)
        loop.add_spare_axpby(tape, e, loop.x, m)
        loop.add_transition(tape, x_t=xp, x_s=loop.x, eps=e)
        return [tape]

    @_lib.on_device
    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        q_t = self.denoiser(x_t, t, **kwargs)
        if not x_t.is_cuda:  # host tensors: reference op sequence
            z_t = (x_t - alpha_t * q_t.mean) / sigma_t
            x_s = alpha_s / alpha_t * x_t + alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t) * z_t
            q_s = self.denoiser(x_s, s, **kwargs)
            z_s = (x_s - alpha_s * q_s.mean) / sigma_s
            z_t = (z_t + z_s) / 2
            return alpha_s / alpha_t * x_t + alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t) * z_t
        c = alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t)
        z_t = _lin2(1 / sigma_t, x_t, -alpha_t / sigma_t, q_t.mean)
        x_p = _lin2(alpha_s / alpha_t, x_t, c, z_t)
        q_s = self.denoiser(x_p, s, **kwargs)
        z_s = _lin2(1 / sigma_s, x_p, -alpha_s / sigma_s, q_s.mean)
        z_m = _lin2(torch.full_like(c, 0.5), z_t, torch.full_like(c, 0.5), z_s)
        return _lin2(alpha_s / alpha_t, x_t, c, z_m)


class ItoSampler(Sampler):
    r"""Ito SDE sampler (reference ``azula/sample.py:355-431``):
    x_s = alpha_s/alpha_t x_t + (1 + eta^2)/tau (sigma_s/sigma_t - alpha_s/alpha_t)(x_t - alpha_t mu)
          + eta alpha_s sqrt|sigma_t^2/alpha_t^2 - sigma_s^2/alpha_s^2| eps  -- the fused kernel's form."""

    def __init__(self, denoiser: Denoiser, eta: float = 1.0, temperature: float = 1.0, **kwargs) -> None:
        super().__init__(**kwargs)
        self.denoiser = denoiser
        self.eta = eta
        self.temperature = temperature

    _noise_like_update = True  # randn_like(x_s), reference azula/sample.py:427-429

    def _ito(self, alpha_t, sigma_t, alpha_s, sigma_s):
        k = (1 + self.eta**2) / self.temperature * (sigma_s / sigma_t - alpha_s / alpha_t)
        k_eps = self.eta * alpha_s * torch.sqrt(torch.abs((sigma_t / alpha_t) ** 2 - (sigma_s / alpha_s) ** 2))
        return alpha_s / alpha_t, k, k_eps

    def _kernel_scalars(self, alpha_t, sigma_t, alpha_s, sigma_s):
        r_, k, k_eps = self._ito(alpha_t, sigma_t, alpha_s, sigma_s)
        # x_s = r x + k (x - a_t m) = (-k a_t) m + (r + k) (x - 0 m)
        return torch.zeros_like(k), -k * alpha_t, r_ + k, k_eps

    def _needs_noise(self) -> bool:
        return self.eta != 0

    def _host_step(self, x_t, mean, t, s):
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        r_, k, k_eps = self._ito(alpha_t, sigma_t, alpha_s, sigma_s)
        x_s = r_ * x_t
        x_s = x_s + k * (x_t - alpha_t * mean)
        return x_s + k_eps * self._draw_noise(x_s)

    @_lib.on_device
    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        return self._step_impl(x_t, t, s, **kwargs)


# ------------------------------------------------------------------------------- multistep family
def _lagrange_weights(u: Tensor, i: int, n: int, moments: Callable[[Tensor, int, Tensor], Tensor]) -> Tensor:
    r"""Weights of the order-``n`` linear multistep rule on the nodes ``u[i+1-n : i+1]``: solve the
    Vandermonde system ``sum_j c_j u_j^k = moments_k`` in fp64 and round back to ``u``'s dtype, as the
    reference's ``promote_dtype(min_dtype=float64)`` static methods do (``azula/sample.py:486-508``)."""
    wide = u.to(torch.promote_types(u.dtype, torch.float64))
    n = min(n, i + 1)
    k = torch.arange(n, device=u.device)
    V = wide[i + 1 - n : i + 1] ** k[:, None]
    return torch.linalg.solve(V, moments(wide, i, k)).to(u.dtype)


def _moments_poly(t: Tensor, i: int, k: Tensor) -> Tensor:
    r"""int_{t_i}^{t_{i+1}} u^k du (``azula/sample.py:505-506``)."""
    return t[i + 1] ** (k + 1) / (k + 1) - t[i] ** (k + 1) / (k + 1)


def _moments_exp(t: Tensor, i: int, k: Tensor) -> Tensor:
    r"""int e^u u^k du by repeated integration by parts (``azula/sample.py:670-683``)."""
    k_fact = torch.cumprod(torch.clip(k, min=1), dim=0)
    hi = torch.exp(t[i + 1]) * torch.cumsum((-t[i + 1]) ** k / k_fact, dim=0)
    lo = torch.exp(t[i]) * torch.cumsum((-t[i]) ** k / k_fact, dim=0)
    return (-1) ** k * k_fact * (hi - lo)


def _moments_negexp(t: Tensor, i: int, k: Tensor) -> Tensor:
    r"""int e^{-u} u^k du (``azula/sample.py:783-792``)."""
    k_fact = torch.cumprod(torch.clip(k, min=1), dim=0)
    hi = torch.exp(-t[i + 1]) * torch.cumsum(t[i + 1] ** k / k_fact, dim=0)
    lo = torch.exp(-t[i]) * torch.cumsum(t[i] ** k / k_fact, dim=0)
    return -k_fact * (hi - lo)


def _moments_sech(t: Tensor, i: int, k: Tensor) -> Tensor:
    r"""int e^u / (1 + e^{2u}) u^k du, 256-panel trapezoid (``azula/sample.py:907-910``)."""
    u = torch.linspace(t[i], t[i + 1], steps=256 + 1, dtype=t.dtype, device=t.device)
    y = torch.exp(u) / (1 + torch.exp(2 * u)) * (u ** k[:, None])
    return torch.trapezoid(y, u, dim=-1)


class _MultistepSampler(Sampler):
    r"""Shared loop of the Adams-Bashforth family.  A subclass states, in the reference's op order,

    * ``_variable(alpha, sigma)``: the integration variable u_t,
    * ``_predict(x_t, mean, alpha_t, sigma_t)``: the buffered prediction (z, v, x or f),
    * ``_advance(x_t, integral, alpha_t, sigma_t, alpha_s, sigma_s)``: x_s,
    * ``_moments``: the basis integrals of its rule.

    Host tensors run exactly that sequence.  Device tensors use that both maps are LINEAR: probing
    them with 0-d fp64 unit inputs yields (a, b) and (p, q) of ``az_multistep_f32``
    (pred = a x_t + b mean, x_s = p x_t + q sum_j c_j pred_j), so each step is ONE pass over
    2 + (order - 1) streams instead of ~3 order + 6 ATen passes.  The whole (steps, 4 + order)
    coefficient table is built on the host (fp64 solves) and uploaded once."""

    _moments: Callable[[Tensor, int, Tensor], Tensor]

    def __init__(self, denoiser: Denoiser, order: int = 2, **kwargs) -> None:
        super().__init__(**kwargs)
        self.denoiser = denoiser
        self.order = order

    def _variable(self, alpha: Tensor, sigma: Tensor) -> Tensor:
        raise NotImplementedError()

    def _predict(self, x_t, mean, alpha_t, sigma_t):
        raise NotImplementedError()

    def _advance(self, x_t, integral, alpha_t, sigma_t, alpha_s, sigma_s):
        raise NotImplementedError()

    @classmethod
    def _weights(cls, u: Tensor, i: int, n: int) -> Tensor:
        return _lagrange_weights(u, i, n, cls._moments)

    def _device_table(self, alpha: Tensor, sigma: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
        r"""(steps, 4 + MAX_HIST) rows [a, b, p, w_new, w_hist oldest-first] from host scalars (fp32 for the kernel)."""
        u = self._variable(alpha, sigma)
        rows = torch.zeros(self.steps, 4 + _lib.MULTISTEP_MAX_HIST, dtype=dtype)
        one, zero = torch.ones((), dtype=torch.float64), torch.zeros((), dtype=torch.float64)
        al, sg = alpha.double(), sigma.double()
        for i in range(self.steps):
            c = self._weights(u, i, self.order).double()
            rows[i, 0] = self._predict(one, zero, al[i], sg[i])
            rows[i, 1] = self._predict(zero, one, al[i], sg[i])
            rows[i, 2] = self._advance(one, zero, al[i], sg[i], al[i + 1], sg[i + 1])
            q = self._advance(zero, one, al[i], sg[i], al[i + 1], sg[i + 1])
            rows[i, 3] = q * c[-1]
            rows[i, 4 : 4 + len(c) - 1] = q * c[:-1]
        return rows

    # -- fused form: the history ring makes the buffer addresses cycle with period `order`, so ONE graph holds `order`
    # consecutive steps (ring slot = step mod order, fixed per phase) and is replayed steps / order times (a second,
    # shorter graph covers the remainder).  Every step uses the full-length kernel (order - 1 history streams); the
    # warm-up steps simply carry zero weights for the entries that do not exist yet (the ring is zeroed per call).
    def _needs_noise(self) -> bool:
        return False

    def _fused_structure(self) -> tuple:
        return (*super()._fused_structure(), self.order)

    def _fused_rows(self, t, s, fused):
        alpha_t, sigma_t = self.denoiser.schedule(t)
        zero = torch.zeros_like(alpha_t)
        return [dict(alpha=alpha_t, sigma=sigma_t, a_t=zero, a_s=zero, k_x=zero, k_eps=zero)]  # only the mean is used

    def _ring_table(self) -> Tensor:
        r"""``_device_table`` with the history weights RIGHT-aligned over order - 1 slots (slot k <-> step i - (order-1) + k)."""
        alpha, sigma = self.denoiser.schedule(torch.linspace(self.start, self.stop, self.steps + 1, dtype=self.dtype))
        rows = self._device_table(alpha, sigma)
        nh = self.order - 1
        out = torch.zeros_like(rows)
        out[:, :4] = rows[:, :4]
        for i in range(self.steps):
            n = min(self.order, i + 1) - 1
            out[i, 4 + nh - n : 4 + nh] = rows[i, 4 : 4 + n]
        return out

    def _fused_step_tapes(self, loop):
        if self.order < 1 or self.order > _lib.MULTISTEP_MAX_HIST + 1:
            raise ValueError(f"order must be in [1, {_lib.MULTISTEP_MAX_HIST + 1}], got {self.order}")
        dev, nh = loop.x.device, self.order - 1
        ncols = 4 + _lib.MULTISTEP_MAX_HIST
        loop.ring = [torch.zeros_like(loop.x) for _ in range(self.order)]
        if isinstance(loop, _FusedLoopWide):
            # fp64 clock: the statements of `_call_wide` (pred = a x_t + b mean; acc = p x_t + w_new pred; acc += w_j pred_j, oldest
            # first) as az_axpby_f64 launches reading words 36 .. 47 of the current fp64 row: [a, b, p, w_new, w_hist x 7, 1]
            mean, scratch = torch.empty_like(loop.x), torch.empty_like(loop.x)
            loop.keep += [mean, scratch]
            w = lambda k: loop.cur64.data_ptr() + 8 * (36 + k)  # noqa: E731
            n = loop.x.numel()
            tapes = []
            for j in range(self.order):
                tape = Tape()
                loop._src = loop.x
                loop.add_evaluation(tape)
                loop.add_mean(tape, loop.x, mean)
                pred = loop.ring[j]
                tape.add("az_axpby_f64", pred.data_ptr(), w(0), loop.x.data_ptr(), w(1), mean.data_ptr(), 0, 1, n, 0)
                acc = scratch if nh else loop.x
                tape.add("az_axpby_f64", acc.data_ptr(), w(2), loop.x.data_ptr(), w(3), pred.data_ptr(), 0, 1, n, 0)
                for k in range(nh):  # slot k <-> step i - nh + k (right-aligned weights: zero while that step does not exist)
                    dst = loop.x if k == nh - 1 else scratch
                    tape.add("az_axpby_f64", dst.data_ptr(), w(11), scratch.data_ptr(), w(4 + k),
                             loop.ring[(j - nh + k) % self.order].data_ptr(), 0, 1, n, 0)
                tapes.append(tape)
            return tapes
        loop.mtable = torch.zeros(self.steps, ncols, dtype=torch.float32, device=dev)
        mrow = torch.zeros(ncols, dtype=torch.float32, device=dev)
        mean, scratch = torch.empty_like(loop.x), torch.empty_like(loop.x)
        loop.keep += [mrow, mean, scratch]
        tapes = []
        for j in range(self.order):
            tape = Tape()
            loop.add_evaluation(tape)
            # posterior mean (preconditioning, clip, CFG combine) through the transition kernel's mean output
            loop.add_transition(tape, x_t=loop.x, x_s=scratch, mean_out=mean, write_xin=False)
            tape.add("az_gather_step_row_f32", mrow.data_ptr(), loop.mtable.data_ptr(), loop.cur.data_ptr(), 1, ncols, self.steps)
            args = _lib.AzMultistepArgs()
            args.x_s, args.x_t, args.mean = loop.x.data_ptr(), loop.x.data_ptr(), mean.data_ptr()
            args.pred = loop.ring[j].data_ptr()
            for k in range(nh):  # oldest first: steps i - nh .. i - 1 live in slots (j - nh + k) mod order
                args.hist[k] = loop.ring[(j - nh + k) % self.order].data_ptr()
            args.coef, args.count, args.n_hist = mrow.data_ptr(), loop.x.numel(), nh
            tape.add("az_multistep_f32", C.byref(args), keep=[args])
            loop.add_input_relayout(tape, loop.x, loop.coef_ptr("c_in_next"))
            tapes.append(tape)
        return tapes

    def _call_wide(self, x: Tensor, time: Tensor, kwargs: dict) -> Tensor:
        r"""fp64 latents / an fp64 time grid (the reference's promotion makes every elementwise op of the loop fp64 around
        the fp32 backbone, ``sample.py:69-94``): the same linear form as the fp32 kernel -- pred = a x_t + b mean,
        x_s = p x_t + w_new pred + sum_j w_j pred_j -- evaluated with ``az_axpby_f64`` from an fp64 coefficient table."""
        alpha, sigma = self.denoiser.schedule(self.timesteps.cpu())
        table = self._device_table(alpha, sigma, dtype=torch.float64)
        ring: list = [None] * self.order
        x_t = x.to(torch.float64).contiguous()
        one = torch.ones((), dtype=torch.float64)
        for i, t in enumerate(self.progress_bar(time[:-1])):
            mean = self.denoiser(x_t, t, **kwargs).mean
            row = table[i]
            pred = axpby_wide(row[0], x_t, row[1], mean)
            n_hist = min(self.order, i + 1) - 1
            acc = axpby_wide(row[2], x_t, row[3], pred)
            for j in range(n_hist):  # oldest first: steps i - n_hist .. i - 1
                acc = axpby_wide(one, acc, row[4 + j], ring[(i - n_hist + j) % self.order])
            ring[i % self.order] = pred
            x_t = acc
        return x_t

    def _fused_upload_extra(self, loop) -> None:
        if not isinstance(loop, _FusedLoopWide):
            loop.mtable.copy_(self._ring_table())

    def _host_table_wide(self, fused: "FusedDenoiser") -> Tensor:
        rows = super()._host_table_wide(fused)
        alpha, sigma = self.denoiser.schedule(torch.linspace(self.start, self.stop, self.steps + 1, dtype=self.dtype))
        tab = self._device_table(alpha, sigma, dtype=torch.float64)
        nh = self.order - 1
        rows[:, 36:40] = tab[:, :4]
        for i in range(self.steps):
            n = min(self.order, i + 1) - 1
            rows[i, 40 + nh - n : 40 + nh] = tab[i, 4 : 4 + n]
        rows[:, 47] = 1.0
        return rows

    def _fused_reset(self, loop) -> None:
        for r in loop.ring:
            r.zero_()

    @torch.no_grad()
    @_lib.on_device
    def __call__(self, x: Tensor, **kwargs) -> Tensor:
        if self.order < 1 or self.order > _lib.MULTISTEP_MAX_HIST + 1:
            raise ValueError(f"order must be in [1, {_lib.MULTISTEP_MAX_HIST + 1}], got {self.order}")
        time = self.timesteps.to(device=x.device)
        if not x.is_cuda:
            return self._call_host(x, time, kwargs)
        if type(self).__call__ is _MultistepSampler.__call__ and x.ndim >= 2 and (
                (x.dtype == torch.float32 and self.dtype in (None, torch.float32)) or self._wide_fusable(x)):
            out = self._call_fused(x, kwargs)
            if out is not None:
                return out
        if self.dtype == torch.float64 or x.dtype == torch.float64:
            return self._call_wide(x, time, kwargs)
        require_f32_cuda(x, type(self).__name__)
        alpha, sigma = self.denoiser.schedule(self.timesteps.cpu())
        table = self._device_table(alpha, sigma).to(x.device)
        ring = [torch.empty_like(x, memory_format=torch.contiguous_format) for _ in range(self.order)]
        x_t = x.contiguous()
        x_next = [torch.empty_like(x_t), torch.empty_like(x_t)]
        stream = _lib.stream_ptr()
        for i, t in enumerate(self.progress_bar(time[:-1])):
            mean = self.denoiser(x_t, t, **kwargs).mean.to(x_t).contiguous()
            n_hist = min(self.order, i + 1) - 1
            args = _lib.AzMultistepArgs()
            args.x_s = x_next[i % 2].data_ptr()
            args.pred = ring[i % self.order].data_ptr()
            args.x_t, args.mean = x_t.data_ptr(), mean.data_ptr()
            for j in range(n_hist):  # oldest first: steps i - n_hist .. i - 1
                args.hist[j] = ring[(i - n_hist + j) % self.order].data_ptr()
            args.coef = table[i].data_ptr()
            args.count, args.n_hist = x_t.numel(), n_hist
            _lib.call("az_multistep_f32", C.byref(args), stream)
            x_t = x_next[i % 2]
        return x_t

    def _call_host(self, x: Tensor, time: Tensor, kwargs: dict) -> Tensor:
        alpha, sigma = self.denoiser.schedule(time)
        u = self._variable(alpha, sigma)
        x_t, buffer = x, []
        for i, t in enumerate(self.progress_bar(time[:-1])):
            q_t = self.denoiser(x_t, t, **kwargs)
            buffer.append(self._predict(x_t, q_t.mean, alpha[i], sigma[i]))
            if len(buffer) > self.order:
                buffer.pop(0)
            coeffs = self._weights(u, i, self.order)
            integral = sum(b * c for b, c in zip(buffer, coeffs, strict=True))
            x_t = self._advance(x_t, integral, alpha[i], sigma[i], alpha[i + 1], sigma[i + 1])
        return x_t


class zABSampler(_MultistepSampler):
    r"""Adams-Bashforth multistep sampler with noise (z) prediction, u = sigma / alpha
    (reference ``azula/sample.py:434-546``; rho-AB of Zhang et al. 2023, k-diffusion's LMS)."""

    _moments = staticmethod(_moments_poly)

    @staticmethod
    def _adams_bashforth(t: Tensor, /, i: int, n: int) -> Tensor:
        return _lagrange_weights(t, i, n, _moments_poly)

    def _variable(self, alpha, sigma):
        return sigma / alpha

    def _predict(self, x_t, mean, alpha_t, sigma_t):
        return (x_t - alpha_t * mean) / sigma_t

    def _advance(self, x_t, integral, alpha_t, sigma_t, alpha_s, sigma_s):
        return alpha_s / alpha_t * x_t + alpha_s * integral


class vABSampler(zABSampler):
    r"""Adams-Bashforth sampler with velocity (v) prediction, u = sigma / (alpha + sigma)
    (reference ``azula/sample.py:549-620``)."""

    def _variable(self, alpha, sigma):
        return sigma / (alpha + sigma)

    def _predict(self, x_t, mean, alpha_t, sigma_t):
        return 1 / sigma_t * x_t - (1 + alpha_t / sigma_t) * mean

    def _advance(self, x_t, integral, alpha_t, sigma_t, alpha_s, sigma_s):
        return (alpha_s + sigma_s) / (alpha_t + sigma_t) * x_t + (alpha_s + sigma_s) * integral


class zEABSampler(_MultistepSampler):
    r"""Exponential Adams-Bashforth sampler with noise prediction, u = log sigma - log alpha
    (reference ``azula/sample.py:623-715``; DPM-Solver family)."""

    _moments = staticmethod(_moments_exp)

    @staticmethod
    def _exponential_adams_bashforth(t: Tensor, /, i: int, n: int) -> Tensor:
        return _lagrange_weights(t, i, n, _moments_exp)

    def _variable(self, alpha, sigma):
        return sigma.log() - alpha.log()

    def _predict(self, x_t, mean, alpha_t, sigma_t):
        return (x_t - alpha_t * mean) / sigma_t

    def _advance(self, x_t, integral, alpha_t, sigma_t, alpha_s, sigma_s):
        return alpha_s / alpha_t * x_t + alpha_s * integral


class xEABSampler(_MultistepSampler):
    r"""Exponential Adams-Bashforth sampler with data (x) prediction
    (reference ``azula/sample.py:718-821``; DPM-Solver++ family)."""

    _moments = staticmethod(_moments_negexp)

    @staticmethod
    def _exponential_adams_bashforth(t: Tensor, /, i: int, n: int) -> Tensor:
        return _lagrange_weights(t, i, n, _moments_negexp)

    def _variable(self, alpha, sigma):
        return sigma.log() - alpha.log()

    def _predict(self, x_t, mean, alpha_t, sigma_t):
        return mean

    def _advance(self, x_t, integral, alpha_t, sigma_t, alpha_s, sigma_s):
        return sigma_s / sigma_t * x_t - sigma_s * integral


class REABSampler(_MultistepSampler):
    r"""Rescaled exponential Adams-Bashforth sampler (reference ``azula/sample.py:824-950``): the
    prediction f_t = (1 - a_t) / (b_t alpha_t) x_t - mean / b_t with a_t = sigma_t^2 / (alpha_t^2 +
    sigma_t^2), b_t = sqrt(a_t); basis integrals by 256-panel trapezoid."""

    _moments = staticmethod(_moments_sech)

    @staticmethod
    def _exponential_adams_bashforth(t: Tensor, /, i: int, n: int) -> Tensor:
        return _lagrange_weights(t, i, n, _moments_sech)

    def _variable(self, alpha, sigma):
        return sigma.log() - alpha.log()

    def _predict(self, x_t, mean, alpha_t, sigma_t):
        a_t = sigma_t**2 / (alpha_t**2 + sigma_t**2)
        b_t = sigma_t * torch.rsqrt(alpha_t**2 + sigma_t**2)
        return (1 - a_t) / b_t / alpha_t * x_t - 1 / b_t * mean

    def _advance(self, x_t, integral, alpha_t, sigma_t, alpha_s, sigma_s):
        return (
            torch.sqrt((alpha_s**2 + sigma_s**2) / (alpha_t**2 + sigma_t**2)) * x_t
            + torch.sqrt(alpha_s**2 + sigma_t**2) * integral
        )


class PCSampler(Sampler):
    r"""Predictor-corrector sampler (reference ``azula/sample.py:953-993``): ``corrections`` Langevin-like
    corrector moves at time t, then a deterministic (DDIM eta=0) predictor to s.  Both are instances
    of the fused transition kernel's form x' = a_s m + k_x (x - a_t m) + k_eps eps."""

    def __init__(self, denoiser: Denoiser, corrections: int = 1, delta: float = 0.01, **kwargs) -> None:
        super().__init__(**kwargs)
        self.denoiser = denoiser
        self.corrections = corrections
        self.delta = delta

    # Fused form: corrections + 1 table rows / denoiser evaluations per step, all the kernel's own form
    # x' = a_s m + k_x (x - a_t m) + k_eps eps, in place; one noise buffer per corrector move, drawn before the replay
    # in the reference's order (its predictor draws nothing).
    def _needs_noise(self) -> bool:
        return self.corrections > 0

    def _noise_draws(self) -> int:
        return self.corrections

    def _fused_structure(self) -> tuple:
        return (*super()._fused_structure(), self.corrections)

    def _fused_rows(self, t, s, fused):
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        keep, kick = math.sqrt(1 - self.delta), math.sqrt(self.delta)
        corr = dict(alpha=alpha_t, sigma=sigma_t, a_t=alpha_t, a_s=alpha_t, k_x=keep + 0 * alpha_t, k_eps=kick * sigma_t)
        pred = dict(alpha=alpha_t, sigma=sigma_t, a_t=alpha_t, a_s=alpha_s, k_x=sigma_s / sigma_t, k_eps=0 * alpha_t)
        return [corr] * self.corrections + [pred]

    def _fused_step_tapes(self, loop):
        tape = Tape()
        for k in range(self.corrections):
            loop.add_evaluation(tape)
            loop.add_transition(tape, x_t=loop.x, x_s=loop.x, eps=loop.noise[k])
        loop.add_evaluation(tape)
        loop.add_transition(tape, x_t=loop.x, x_s=loop.x)
        return [tape]

    @_lib.on_device
    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        keep, kick = math.sqrt(1 - self.delta), math.sqrt(self.delta)
        for _ in range(self.corrections):
            q_t = self.denoiser(x_t, t, **kwargs)
            if x_t.is_cuda:
                x_t = self._device_kernel(x_t, q_t.mean, alpha_t, alpha_t, keep + 0 * alpha_t, kick * sigma_t, True)
            else:
                x_t = alpha_t * q_t.mean + keep * (x_t - alpha_t * q_t.mean) + kick * sigma_t * self._draw_noise(x_t)
        q_t = self.denoiser(x_t, t, **kwargs)
        if x_t.is_cuda:
            return self._device_kernel(x_t, q_t.mean, alpha_t, alpha_s, sigma_s / sigma_t, 0 * alpha_t, False)
        return alpha_s * q_t.mean + sigma_s / sigma_t * (x_t - alpha_t * q_t.mean)


# samplers whose ``step`` the captured loop reproduces (a subclass overriding ``step`` falls back to the generic loop)
_FUSED_STEPS = (DDPMSampler.step, DDIMSampler.step, EulerSampler.step, HeunSampler.step, ItoSampler.step, PCSampler.step)
