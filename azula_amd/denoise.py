r"""Denoisers and posteriors -- drop-in for ``azula.denoise`` on the sampling path.

``Denoiser.forward(x_t, t, **kwargs) -> Posterior`` with ``.mean`` shaped like ``x_t`` and an
attribute/property ``schedule`` (reference ``azula/denoise.py:97-114``).

Device tensors: the preconditioning arithmetic runs in HIP kernels through the C ABI
(``az_scale_f32``, ``az_axpby_f32``); inside a sampler it is fused with the transition
(``az_transition_f32``) and never exists as separate passes.  Host tensors (the reference's
CPU-runnable README configuration) take the reference's own op sequence in torch.
"""

from __future__ import annotations

import abc
import math

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .nn.utils import get_module_dtype
from .noise import Schedule

__all__ = ["Posterior", "DiracPosterior", "GaussianPosterior", "Denoiser", "SimpleDenoiser", "KarrasDenoiser"]


class Posterior(abc.ABC):
    r"""Abstract posterior q(X | x_t) (reference ``azula/denoise.py:50-53``)."""

    mean: Tensor


class DiracPosterior(Posterior):
    r"""Dirac delta at ``mean`` (reference ``azula/denoise.py:56-66``)."""

    def __init__(self, mean: Tensor) -> None:
        self.mean = mean


class GaussianPosterior(Posterior):
    r"""N(mean, var) with elementwise variance (reference ``azula/denoise.py:69-94``)."""

    def __init__(self, mean: Tensor, var: Tensor) -> None:
        self.mean = mean
        self.var = var

    def log_prob(self, x: Tensor) -> Tensor:
        return -((x - self.mean) ** 2 / self.var + torch.log(self.var) + math.log(2 * math.pi)) / 2


class Denoiser(nn.Module):
    r"""Abstract denoiser module (reference ``azula/denoise.py:97-114``)."""

    schedule: Schedule

    @abc.abstractmethod
    def forward(self, x_t: Tensor, t: Tensor, **kwargs) -> Posterior:
        r"""x_t: (B, *), t: () or (B) -> posterior."""

    # -- fused sampling protocol (azula_amd internal) ---------------------------------------------
    def _az_fused(self, x: Tensor, kwargs: dict, cur_coef: Tensor):
        r"""Returns a :class:`azula_amd.sample.FusedDenoiser` if this denoiser can run inside the
        captured per-step graph, else ``None`` (generic step-by-step path)."""
        return None


def _expand_like(a: Tensor, ndim: int) -> Tensor:
    while a.ndim < ndim:
        a = a[..., None]
    return a


def require_f32_cuda(x: Tensor, who: str) -> None:
    if x.dtype != torch.float32:
        raise NotImplementedError(
            f"{who}: the gfx950 kernels are fp32; got a {x.dtype} device tensor and there is no eager fallback"
        )


def karras_coefficients(alpha_t: Tensor, sigma_t: Tensor):
    r"""(c_in, c_out, c_skip, c_time) in the reference's op order (``azula/denoise.py:309-312``)."""
    c_in = torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_out = sigma_t * torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_skip = alpha_t / (alpha_t**2 + sigma_t**2)
    c_time = torch.log(sigma_t / alpha_t)
    return c_in, c_out, c_skip, c_time


def is_wide(*tensors: Tensor) -> bool:
    r"""True if torch's type promotion makes the elementwise path fp64: a DIMENSIONED fp64 tensor takes part (0-d
    tensors do not promote).  ``Sampler(dtype=float64)`` gets there through the schedule scalars, which the denoisers
    expand to shape (1, ..., 1) before multiplying them into x_t (reference ``azula/denoise.py:306-322``)."""
    return any(t.dtype == torch.float64 and t.ndim > 0 for t in tensors)


def _dev64(c: Tensor, device) -> Tensor:
    return c.reshape(-1).to(device=device, dtype=torch.float64).contiguous()


def precondition_wide(x_t: Tensor, c_in: Tensor) -> Tensor:
    r"""``(c_in * x_t).to(float32)`` with the product taken in fp64 (``az_scale_f64_to_f32``)."""
    x64 = x_t.to(torch.float64).contiguous()
    c = _dev64(c_in, x_t.device)
    rows = x_t.shape[0] if c.numel() > 1 else 1
    y = torch.empty(x_t.shape, dtype=torch.float32, device=x_t.device)
    _lib.call("az_scale_f64_to_f32", y.data_ptr(), x64.data_ptr(), c.data_ptr(), rows, x64.numel() // rows,
              1 if c.numel() > 1 else 0, _lib.stream_ptr())
    return y


def axpby_wide(a: Tensor, x: Tensor, b: Tensor, z: Tensor) -> Tensor:
    r"""a * x + b * z in fp64 (``az_axpby_f64``); ``z`` may be fp32 (widened element by element, as torch does)."""
    x64 = x.to(torch.float64).contiguous()
    z = z.contiguous() if z.dtype in (torch.float32, torch.float64) else z.to(torch.float32).contiguous()
    a64, b64 = _dev64(a, x.device), _dev64(b, x.device)
    rows = x.shape[0] if a64.numel() > 1 else 1
    y = torch.empty_like(x64)
    _lib.call("az_axpby_f64", y.data_ptr(), a64.data_ptr(), x64.data_ptr(), b64.data_ptr(), z.data_ptr(),
              int(z.dtype == torch.float32), rows, x64.numel() // rows, 1 if a64.numel() > 1 else 0, _lib.stream_ptr())
    return y


def precondition(x_t: Tensor, c_in: Tensor) -> Tensor:
    r"""c_in * x_t on the device (reference ``azula/denoise.py:317``); c_in is () or (B,)."""
    B = x_t.shape[0] if c_in.numel() > 1 else 1
    y = torch.empty_like(x_t)
    c = c_in.reshape(-1).to(torch.float32).contiguous()
    zero = torch.zeros_like(c)
    # y = c * x + 0 * x  (exact: adding a signed zero leaves every finite value unchanged)
    _lib.call(
        "az_axpby_f32", y.data_ptr(), c.data_ptr(), x_t.data_ptr(), zero.data_ptr(), x_t.data_ptr(),
        B, x_t.numel() // B, 1 if c.numel() > 1 else 0, _lib.stream_ptr(),
    )
    return y


def postcondition(x_t: Tensor, out: Tensor, c_skip: Tensor, c_out: Tensor) -> Tensor:
    r"""c_skip * x_t + c_out * out on the device (reference ``azula/denoise.py:322``)."""
    B = x_t.shape[0] if c_skip.numel() > 1 else 1
    y = torch.empty_like(x_t)
    a = c_skip.reshape(-1).to(torch.float32).contiguous()
    b = c_out.reshape(-1).to(torch.float32).contiguous()
    _lib.call(
        "az_axpby_f32", y.data_ptr(), a.data_ptr(), x_t.data_ptr(), b.data_ptr(), out.data_ptr(),
        B, x_t.numel() // B, 1 if a.numel() > 1 else 0, _lib.stream_ptr(),
    )
    return y


class KarrasDenoiser(Denoiser):
    r"""EDM-style preconditioned denoiser (reference ``azula/denoise.py:263-324``).

    mu(x_t) = c_skip x_t + c_out F(c_in x_t, c_time),  with
    c_in = rsqrt(a^2 + s^2), c_out = s c_in, c_skip = a / (a^2 + s^2), c_time = log(s / a).
    """

    def __init__(self, backbone: nn.Module, schedule: Schedule) -> None:
        super().__init__()
        self.backbone = backbone
        self.schedule = schedule

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x_t: Tensor, t: Tensor, **kwargs) -> DiracPosterior:
        alpha_t, sigma_t = self.schedule(t)
        alpha_t, sigma_t = _expand_like(alpha_t, x_t.ndim), _expand_like(sigma_t, x_t.ndim)
        c_in, c_out, c_skip, c_time = karras_coefficients(alpha_t, sigma_t)
        c_time = c_time.reshape_as(t)
        dtype = get_module_dtype(self.backbone) or x_t.dtype

        if not x_t.is_cuda:  # host tensors: the reference's op sequence (README CPU configuration)
            output = self.backbone((c_in * x_t).to(dtype), c_time.to(dtype), **kwargs).to(x_t)
            return DiracPosterior(mean=c_skip * x_t + c_out * output)

        if is_wide(x_t, alpha_t):  # fp64 time grid and / or fp64 latents: the elementwise path is fp64 (see is_wide)
            x_in = precondition_wide(x_t, c_in)
            output = self.backbone(x_in.to(dtype), c_time.to(device=x_t.device, dtype=dtype), **kwargs)
            output = output.to(x_t.dtype)  # .to(x_t): the latents' dtype, before the fp64 scalars promote the sum
            return DiracPosterior(mean=axpby_wide(c_skip, x_t, c_out, output))
        require_f32_cuda(x_t, "KarrasDenoiser")
        x_t = x_t.contiguous()
        x_in = precondition(x_t, c_in.to(x_t.device))
        output = self.backbone(x_in.to(dtype), c_time.to(device=x_t.device, dtype=dtype), **kwargs)
        output = output.to(x_t).contiguous()
        return DiracPosterior(mean=postcondition(x_t, output, c_skip.to(x_t.device), c_out.to(x_t.device)))

    # -- fused sampling -------------------------------------------------------------------------
    def host_coefficients(self, alpha_t: Tensor, sigma_t: Tensor) -> dict:
        r"""Per-step scalars for the device table, from 0-d HOST tensors (reference op order)."""
        c_in, c_out, c_skip, c_time = karras_coefficients(alpha_t, sigma_t)
        return {"c_in": c_in, "c_out": c_out, "c_skip": c_skip, "c_time": c_time}

    def _az_fused(self, x: Tensor, kwargs: dict, cur_coef: Tensor):
        compile_ = getattr(self.backbone, "_az_compile", None)
        if compile_ is None or get_module_dtype(self.backbone) not in (None, torch.float32, torch.float16, torch.bfloat16):
            return None
        program = compile_(x, kwargs, cur_coef)
        if program is None:
            return None
        from .sample import FusedDenoiser

        return FusedDenoiser(coefficients=self.host_coefficients, programs=[program])


class SimpleDenoiser(Denoiser):
    r"""Denoiser whose backbone predicts the mean directly (reference ``azula/denoise.py:177-230``):
    mu(x_t) = F(c_in x_t, c_time) with c_in = rsqrt(a^2 + s^2), c_time = log(s / a).  In the fused step it
    is the Karras form with c_skip = 0, c_out = 1 (0 * x + 1 * F is exact in fp32)."""

    def __init__(self, backbone: nn.Module, schedule: Schedule) -> None:
        super().__init__()
        self.backbone = backbone
        self.schedule = schedule

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x_t: Tensor, t: Tensor, **kwargs) -> DiracPosterior:
        alpha_t, sigma_t = self.schedule(t)
        alpha_t, sigma_t = _expand_like(alpha_t, x_t.ndim), _expand_like(sigma_t, x_t.ndim)
        c_in = torch.rsqrt(alpha_t**2 + sigma_t**2)
        c_time = torch.log(sigma_t / alpha_t).reshape_as(t)
        dtype = get_module_dtype(self.backbone) or x_t.dtype
        if not x_t.is_cuda:
            return DiracPosterior(mean=self.backbone((c_in * x_t).to(dtype), c_time.to(dtype), **kwargs).to(x_t))
        if is_wide(x_t, alpha_t):
            x_in = precondition_wide(x_t, c_in)
            output = self.backbone(x_in.to(dtype), c_time.to(device=x_t.device, dtype=dtype), **kwargs)
            return DiracPosterior(mean=output.to(x_t))
        require_f32_cuda(x_t, "SimpleDenoiser")
        x_in = precondition(x_t.contiguous(), c_in.to(x_t.device))
        output = self.backbone(x_in.to(dtype), c_time.to(device=x_t.device, dtype=dtype), **kwargs)
        return DiracPosterior(mean=output.to(x_t))

    def host_coefficients(self, alpha_t: Tensor, sigma_t: Tensor) -> dict:
        c_in = torch.rsqrt(alpha_t**2 + sigma_t**2)
        return {"c_in": c_in, "c_out": torch.ones_like(c_in), "c_skip": torch.zeros_like(c_in),
                "c_time": torch.log(sigma_t / alpha_t)}

    _az_fused = KarrasDenoiser._az_fused
