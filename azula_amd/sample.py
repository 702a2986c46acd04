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
from collections.abc import Iterable, Sequence
from dataclasses import dataclass, field
from typing import Callable

import torch
from torch import Tensor

from . import _lib
from ._lib import COEF_FIELDS, COEF_WORDS
from .denoise import Denoiser, require_f32_cuda
from .engine import StepGraph, Tape, transition_args

__all__ = ["Sampler", "DDPMSampler", "DDIMSampler", "FusedDenoiser", "BackboneProgram"]


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
        single-device random stream sample for sample."""
        if self.shard is None:
            return torch.randn_like(like) if out is None else out.normal_()
        rank, world = self.shard
        full = torch.randn((world * like.shape[0], *like.shape[1:]), dtype=like.dtype, device=like.device)
        mine = full[rank * like.shape[0] : (rank + 1) * like.shape[0]]
        return mine.contiguous() if out is None else out.copy_(mine)

    # ---------------------------------------------------------------------------- shared DDPM/DDIM
    def _tau(self, alpha_t, sigma_t, alpha_s, sigma_s) -> Tensor:
        raise NotImplementedError()

    def _transition_scalars(self, t: Tensor, s: Tensor):
        r"""0-d scalars of one transition in the reference's op order (``azula/sample.py:249-259``)."""
        alpha_s, sigma_s = self.denoiser.schedule(s)
        alpha_t, sigma_t = self.denoiser.schedule(t)
        tau = self._tau(alpha_t, sigma_t, alpha_s, sigma_s)
        k_x = sigma_s * torch.sqrt(1 - tau) / sigma_t
        k_eps = sigma_s * torch.sqrt(tau)
        return alpha_t, sigma_t, alpha_s, sigma_s, k_x, k_eps

    def _step_impl(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        alpha_t, sigma_t, alpha_s, sigma_s, k_x, k_eps = self._transition_scalars(t, s)
        q_t = self.denoiser(x_t, t, **kwargs)
        if not x_t.is_cuda:  # host tensors: reference op sequence
            x_s = alpha_s * q_t.mean
            x_s = x_s + k_x * (x_t - alpha_t * q_t.mean)
            x_s = x_s + k_eps * self._draw_noise(x_t)
            return x_s
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
        mean = q_t.mean.to(x_c).contiguous()
        eps = self._draw_noise(x_c)
        x_s = torch.empty_like(x_c)
        a = transition_args(
            x_t=x_c.data_ptr(), F=mean.data_ptr(), eps=eps.data_ptr(), x_s=x_s.data_ptr(), batch=1, channels=1,
            inner=x_c.numel(), f_channels=1, coef=row.data_ptr(),
        )
        _lib.call("az_transition_f32", C.byref(a), _lib.stream_ptr())
        return x_s

    # ---------------------------------------------------------------------------- fused path
    def _fusable(self, x: Tensor) -> bool:
        cls_step = type(self).step
        if cls_step not in (DDPMSampler.step, DDIMSampler.step):
            return False  # user subclass overrides step (guidance samplers): generic loop
        return x.dtype == torch.float32 and self.dtype in (None, torch.float32) and x.ndim >= 2

    def _host_table(self, fused: FusedDenoiser) -> Tensor:
        r"""(steps, 16) fp32 table of AzStepCoef rows, from 0-d CPU tensors in reference op order."""
        ts = torch.linspace(self.start, self.stop, self.steps + 1, dtype=self.dtype)
        rows = torch.zeros(self.steps, COEF_WORDS, dtype=torch.float32)
        irows = rows.view(torch.int32)  # integer slots (time_index, step) are bit-cast in place
        col = {n: i for i, n in enumerate(COEF_FIELDS)}
        for i, (t, s) in enumerate(ts.unfold(0, 2, 1).unbind()):
            alpha_t, sigma_t, alpha_s, sigma_s, k_x, k_eps = self._transition_scalars(t, s)
            co = fused.coefficients(alpha_t, sigma_t)
            for name, v in co.items():
                if name == "time_index":
                    irows[i, col[name]] = int(v)
                else:
                    rows[i, col[name]] = v.to(torch.float32)
            rows[i, col["alpha_t"]], rows[i, col["alpha_s"]] = alpha_t, alpha_s
            rows[i, col["k_x"]], rows[i, col["k_eps"]] = k_x, k_eps
            rows[i, col["clip_lo"]], rows[i, col["clip_hi"]] = fused.clip
            rows[i, col["guidance"]] = fused.guidance
            irows[i, col["step"]] = i
        rows[:-1, col["c_in_next"]] = rows[1:, col["c_in"]]
        return rows

    def _needs_noise(self) -> bool:
        raise NotImplementedError()

    def _call_fused(self, x: Tensor, kwargs: dict) -> Tensor | None:
        dev = x.device
        g = kwargs.get("guidance")
        key = (
            tuple(x.shape), str(dev), tuple(sorted(kwargs)), self.start, self.stop, self.steps, getattr(self, "eta", None),
            None if torch.is_tensor(g) else g, id(self.denoiser),
        )
        ent = self._fused_cache.get(key)
        if ent is None:
            cur = torch.zeros(COEF_WORDS, dtype=torch.float32, device=dev)
            fused = self.denoiser._az_fused(x, kwargs, cur)
            if fused is None:
                return None
            ent = _FusedLoop(self, fused, x, cur)
            self._fused_cache = {key: ent}  # one live plan per sampler keeps HBM use bounded
        return ent.run(x, kwargs)


class _FusedLoop:
    r"""Static buffers + step tape + hipGraph for one (sampler, denoiser, shape)."""

    def __init__(self, sampler: Sampler, fused: FusedDenoiser, x: Tensor, cur: Tensor) -> None:
        self.sampler, self.fused, self.cur = sampler, fused, cur
        dev = x.device
        self.x = torch.empty_like(x, memory_format=torch.contiguous_format)
        # eps is read by the kernel only when some k_eps != 0; with DDIM eta = 0 the draw is still
        # made (into the same buffer) so that the RNG stream matches the reference, but the
        # transition then moves 12 B/element instead of 16.
        self.eps = torch.empty_like(self.x) if (sampler._needs_noise() or sampler.rng_parity) else None
        self.use_eps = sampler._needs_noise()
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.table_key = None
        self.table = torch.zeros(sampler.steps, COEF_WORDS, dtype=torch.float32, device=dev)
        p0 = fused.programs[0]
        B = x.shape[0]
        Cc = x.shape[1] if x.ndim > 2 else 1
        inner = x.numel() // (B * Cc)
        tape = Tape()
        tape.add("az_step_begin", cur.data_ptr(), self.table.data_ptr(), self.counter.data_ptr(), sampler.steps)
        for p in fused.programs:
            tape.extend(p.tape)
        a = transition_args(
            x_t=self.x.data_ptr(), F=p0.out.data_ptr(),
            F_neg=fused.programs[1].out.data_ptr() if len(fused.programs) > 1 else None,
            eps=self.eps.data_ptr() if self.use_eps else None,
            x_s=self.x.data_ptr(), xin_next=p0.x_in.data_ptr(), batch=B, channels=Cc, inner=inner,
            f_channels=p0.f_channels, f_nhwc=int(p0.f_nhwc), nhwc_pad=p0.x_in_cs, coef=cur.data_ptr(),
        )
        tape.add("az_transition_f32", C.byref(a), keep=[a])
        self.tape = tape
        self.graph: StepGraph | None = None
        self.B, self.C, self.inner = B, Cc, inner

    def _upload_table(self) -> None:
        s = self.sampler
        key = (s.start, s.stop, s.steps, s.dtype, getattr(s, "eta", None), id(s.denoiser.schedule),
               tuple(sorted(vars(s.denoiser.schedule).items())) if hasattr(s.denoiser.schedule, "__dict__") else None,
               self.fused.guidance)
        if key != self.table_key:
            self.table.copy_(s._host_table(self.fused))
            self.table_key = key

    def run(self, x: Tensor, kwargs: dict) -> Tensor:
        s, p0 = self.sampler, self.fused.programs[0]
        self._upload_table()
        for p in self.fused.programs:
            if p.prepare is not None:
                p.prepare(kwargs)
        self.x.copy_(x)
        self.counter.zero_()
        stream = _lib.stream_ptr()
        # backbone input of step 0: c_in[0] * x_T, in the backbone's layout
        c_in0 = self.table[0, COEF_FIELDS.index("c_in")]
        if p0.x_in_cs > 0:
            _lib.call(
                "az_nchw_to_nhwc_f32", p0.x_in.data_ptr(), self.x.data_ptr(), c_in0.data_ptr(), self.B, self.C,
                self.inner, p0.x_in_cs, stream,
            )
        else:
            _lib.call("az_scale_f32", p0.x_in.data_ptr(), self.x.data_ptr(), c_in0.data_ptr(), self.x.numel(), stream)
        for i in s.progress_bar(range(s.steps)):
            if self.eps is not None:
                s._draw_noise(self.eps, out=self.eps)  # same generator calls as the reference's randn_like(x_t)
            if i == 0 and self.graph is None:
                self.tape.run(stream)  # first step eagerly (loads code objects), then capture
                self.graph = StepGraph(self.tape, x.device)
            else:
                self.graph.launch()
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

    def step(self, x_t: Tensor, t: Tensor, s: Tensor, **kwargs) -> Tensor:
        return self._step_impl(x_t, t, s, **kwargs)
