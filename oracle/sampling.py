r"""Oracle: noise schedule, preconditioning and DDPM/DDIM transitions (torch CPU, fp32).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Functional restatement: the
reference's classes become plain functions over tensors so that each formula can be pinned
on its own.
"""

from __future__ import annotations

import math
import torch

from torch import Tensor
from typing import Callable


# --------------------------------------------------------------------------- schedule
def vp_alpha(t: Tensor, alpha_min: float = 1e-3) -> Tensor:
    r"""alpha_t = exp(log(alpha_min) * t^2)  -- azula/noise.py:125-126."""
    return torch.exp(math.log(alpha_min) * t**2)


def vp_sigma(t: Tensor, alpha_min: float = 1e-3, sigma_min: float = 1e-3) -> Tensor:
    r"""sigma_t = sqrt(1 - alpha_t^2 + sigma_min^2)  -- azula/noise.py:128-129."""
    return torch.sqrt(1 - vp_alpha(t, alpha_min) ** 2 + sigma_min**2)


def vp_schedule(t: Tensor, alpha_min: float = 1e-3, sigma_min: float = 1e-3):
    r"""(alpha_t, sigma_t)  -- azula/noise.py:122-123."""
    return vp_alpha(t, alpha_min), vp_sigma(t, alpha_min, sigma_min)


def timesteps(start: float = 1.0, stop: float = 0.0, steps: int = 64, dtype=None) -> Tensor:
    r"""linspace(start, stop, steps + 1)  -- azula/sample.py:86-94."""
    return torch.linspace(start, stop, steps + 1, dtype=dtype)


def time_pairs(start: float = 1.0, stop: float = 0.0, steps: int = 64, dtype=None) -> Tensor:
    r"""(steps, 2) sliding pairs (t, s)  -- azula/sample.py:151."""
    return timesteps(start, stop, steps, dtype).unfold(0, 2, 1)


# --------------------------------------------------------------------------- denoisers
def _expand(a: Tensor, ndim: int) -> Tensor:
    while a.ndim < ndim:  # azula/denoise.py:306-307
        a = a[..., None]
    return a


def karras_coefficients(alpha_t: Tensor, sigma_t: Tensor):
    r"""c_in, c_out, c_skip, c_time  -- azula/denoise.py:309-312."""
    c_in = torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_out = sigma_t * torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_skip = alpha_t / (alpha_t**2 + sigma_t**2)
    c_time = torch.log(sigma_t / alpha_t)
    return c_in, c_out, c_skip, c_time


def karras_mean(
    backbone: Callable[..., Tensor],
    x_t: Tensor,
    t: Tensor,
    schedule: Callable[[Tensor], tuple[Tensor, Tensor]] = vp_schedule,
    backbone_dtype: torch.dtype = torch.float32,
    **kwargs,
) -> Tensor:
    r"""mu = c_skip x_t + c_out F(c_in x_t, c_time)  -- azula/denoise.py:293-324."""
    alpha_t, sigma_t = schedule(t)
    alpha_t, sigma_t = _expand(alpha_t, x_t.ndim), _expand(sigma_t, x_t.ndim)
    c_in, c_out, c_skip, c_time = karras_coefficients(alpha_t, sigma_t)
    c_time = c_time.reshape_as(t)
    out = backbone((c_in * x_t).to(backbone_dtype), c_time.to(backbone_dtype), **kwargs).to(x_t)
    return c_skip * x_t + c_out * out


def adm_sigmas(discrete_schedule: str = "linear", discrete_steps: int = 1000) -> Tensor:
    r"""Discrete sigma table of the ADM plugin  -- azula/plugins/adm/__init__.py:66-84."""
    if discrete_schedule == "linear":
        beta = torch.linspace(
            0.1 / discrete_steps, 20.0 / discrete_steps, discrete_steps, dtype=torch.float64
        )
    elif discrete_schedule == "cosine":
        t = torch.linspace(0, 1, discrete_steps + 1, dtype=torch.float64)
        alpha_bar = torch.cos((t + 0.008) / 1.008 * torch.pi / 2) ** 2
        beta = torch.clip(1 - alpha_bar[1:] / alpha_bar[:-1], max=0.999)
    else:
        raise ValueError(discrete_schedule)
    return torch.sqrt(1 - torch.cumprod(1 - beta, dim=0)).to(torch.float32)


def adm_coefficients(alpha_t: Tensor, sigma_t: Tensor, sigmas: Tensor):
    r"""c_in, c_out, c_skip, time index, c_var  -- azula/plugins/adm/__init__.py:109-114."""
    c_in = torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_out = -sigma_t / alpha_t
    c_skip = 1 / alpha_t
    c_time = sigma_t * torch.rsqrt(alpha_t**2 + sigma_t**2)
    idx = torch.searchsorted(sigmas, c_time.flatten())
    c_var = sigma_t**2 / (alpha_t**2 + sigma_t**2)
    return c_in, c_out, c_skip, idx, c_var


def adm_posterior(
    backbone: Callable[..., Tensor],
    x_t: Tensor,
    t: Tensor,
    sigmas: Tensor,
    label: Tensor | None = None,
    learn_var: bool = True,
    clip_mean: bool = True,
    alpha_min: float = 1e-2,
    sigma_min: float = 1e-2,
):
    r"""(mean, var) of AblatedDenoiser in eval mode  -- azula/plugins/adm/__init__.py:86-136."""
    alpha_t, sigma_t = vp_schedule(t, alpha_min, sigma_min)
    alpha_t, sigma_t = _expand(alpha_t, x_t.ndim), _expand(sigma_t, x_t.ndim)
    c_in, c_out, c_skip, idx, c_var = adm_coefficients(alpha_t, sigma_t, sigmas)
    out = backbone(c_in * x_t, idx, y=label).to(x_t)
    if learn_var:
        out, log_var = torch.chunk(out, 2, dim=1)
        mean = c_skip * x_t + c_out * out
        var = c_var * torch.exp(log_var)
    else:
        mean = c_skip * x_t + c_out * out
        var = c_var
    if clip_mean:
        mean = torch.clip(mean, min=-1.0, max=1.0)
    return mean, var


def cfg_mean(mean_fn, x_t, t, positive: dict, negative: dict, guidance=1.0, **kwargs) -> Tensor:
    r"""mu+ + g (mu+ - mu-), two sequential calls  -- azula/guidance/cfg.py:60-65."""
    pos = mean_fn(x_t, t, **positive, **kwargs)
    neg = mean_fn(x_t, t, **negative, **kwargs)
    return pos + guidance * (pos - neg)


# --------------------------------------------------------------------------- samplers
def sampler_init(shape, alpha_T: Tensor, sigma_T: Tensor, mean=0.0, var=1.0) -> Tensor:
    r"""x_T = alpha_T mean + sqrt(alpha_T^2 var + sigma_T^2) eps  -- azula/sample.py:121-128."""
    mean_T, std_T = alpha_T * mean, torch.sqrt(alpha_T**2 * var + sigma_T**2)
    mean_T, std_T = mean_T.expand(shape), std_T.expand(shape)
    return mean_T + std_T * torch.randn_like(mean_T)


def transition(
    x_t: Tensor,
    mean: Tensor,
    eps: Tensor,
    alpha_t: Tensor,
    sigma_t: Tensor,
    alpha_s: Tensor,
    sigma_s: Tensor,
    eta: float | None,
) -> Tensor:
    r"""DDIM (``eta`` float, azula/sample.py:248-261) or DDPM (``eta=None``, :204-216) update."""
    tau = 1 - (alpha_t / alpha_s * sigma_s / sigma_t) ** 2
    if eta is not None:
        tau = torch.clip(eta * tau, min=0, max=1)
    x_s = alpha_s * mean
    x_s = x_s + sigma_s * torch.sqrt(1 - tau) / sigma_t * (x_t - alpha_t * mean)
    x_s = x_s + sigma_s * torch.sqrt(tau) * eps
    return x_s


def sample(
    mean_fn: Callable[..., Tensor],
    x: Tensor,
    schedule: Callable[[Tensor], tuple[Tensor, Tensor]] = vp_schedule,
    steps: int = 64,
    eta: float | None = 0.0,
    start: float = 1.0,
    stop: float = 0.0,
    eps_list: list[Tensor] | None = None,
    record_eps: list | None = None,
    dtype: torch.dtype | None = None,
    **kwargs,
) -> Tensor:
    r"""Full reverse loop  -- azula/sample.py:139-161 with step :204-216 / :248-261.

    ``dtype`` is the sampler's ``dtype`` argument (azula/sample.py:69-94): the dtype of the TIME GRID.  With float64 the
    schedule scalars, expanded to (1, ..., 1) by the denoiser, promote the latents to fp64 from the first step on.

    ``mean_fn(x_t, t, **kwargs)`` returns the posterior mean.  One ``randn_like`` per step is
    drawn AFTER the denoiser call (even when tau = 0), exactly as the reference does, unless
    ``eps_list`` supplies the noise.
    """
    x_t = x
    for i, (t, s) in enumerate(time_pairs(start, stop, steps, dtype).unbind()):
        alpha_s, sigma_s = schedule(s)
        alpha_t, sigma_t = schedule(t)
        mean = mean_fn(x_t, t, **kwargs)
        eps = torch.randn_like(x_t) if eps_list is None else eps_list[i]
        if record_eps is not None:
            record_eps.append(eps)
        x_t = transition(x_t, mean, eps, alpha_t, sigma_t, alpha_s, sigma_s, eta)
    return x_t


# --------------------------------------------------------------------------- SURVEY 8f samplers
def euler_update(x_t, mean, alpha_t, sigma_t, alpha_s, sigma_s) -> Tensor:
    r"""EulerSampler.step arithmetic -- azula/sample.py:296-305."""
    z_t = (x_t - alpha_t * mean) / sigma_t
    return alpha_s / alpha_t * x_t + alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t) * z_t


def sample_euler(mean_fn, x, schedule=vp_schedule, steps: int = 64, heun: bool = False, **kwargs) -> Tensor:
    r"""Euler (azula/sample.py:296-305) / Heun (:340-352) loops."""
    x_t = x
    for t, s in time_pairs(steps=steps).unbind():
        alpha_s, sigma_s = schedule(s)
        alpha_t, sigma_t = schedule(t)
        mean = mean_fn(x_t, t, **kwargs)
        z_t = (x_t - alpha_t * mean) / sigma_t
        x_s = alpha_s / alpha_t * x_t + alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t) * z_t
        if heun:
            z_s = (x_s - alpha_s * mean_fn(x_s, s, **kwargs)) / sigma_s
            z_t = (z_t + z_s) / 2
            x_s = alpha_s / alpha_t * x_t + alpha_s * (sigma_s / alpha_s - sigma_t / alpha_t) * z_t
        x_t = x_s
    return x_t


def sample_ito(mean_fn, x, schedule=vp_schedule, steps: int = 64, eta: float = 1.0, temperature: float = 1.0,
               eps_list=None, record_eps=None, **kwargs) -> Tensor:
    r"""ItoSampler loop -- azula/sample.py:417-431."""
    x_t = x
    for i, (t, s) in enumerate(time_pairs(steps=steps).unbind()):
        alpha_s, sigma_s = schedule(s)
        alpha_t, sigma_t = schedule(t)
        mean = mean_fn(x_t, t, **kwargs)
        x_s = alpha_s / alpha_t * x_t
        x_s = x_s + (1 + eta**2) / temperature * (sigma_s / sigma_t - alpha_s / alpha_t) * (x_t - alpha_t * mean)
        eps = torch.randn_like(x_s) if eps_list is None else eps_list[i]
        if record_eps is not None:
            record_eps.append(eps)
        x_t = x_s + eta * alpha_s * torch.sqrt(torch.abs((sigma_t / alpha_t) ** 2 - (sigma_s / alpha_s) ** 2)) * eps
    return x_t


# --------------------------------------------------------------------------- SURVEY 8f multistep family
def _basis_integrals(kind: str, t: Tensor, i: int, k: Tensor) -> Tensor:
    r"""Right-hand side of the Lagrange system for each rule (fp64):
    AB  azula/sample.py:505-506;  zEAB :670-683;  xEAB :783-792;  REAB :907-910."""
    lo, hi = t[i], t[i + 1]
    if kind in ("zAB", "vAB"):
        return hi ** (k + 1) / (k + 1) - lo ** (k + 1) / (k + 1)
    if kind == "REAB":
        u = torch.linspace(lo, hi, steps=256 + 1, dtype=t.dtype)
        return torch.trapezoid(torch.exp(u) / (1 + torch.exp(2 * u)) * (u ** k[:, None]), u, dim=-1)
    k_fact = torch.cumprod(torch.clip(k, min=1), dim=0)
    if kind == "zEAB":
        return (-1) ** k * k_fact * (
            torch.exp(hi) * torch.cumsum((-hi) ** k / k_fact, dim=0)
            - torch.exp(lo) * torch.cumsum((-lo) ** k / k_fact, dim=0)
        )
    if kind == "xEAB":
        return -k_fact * (
            torch.exp(-hi) * torch.cumsum(hi**k / k_fact, dim=0) - torch.exp(-lo) * torch.cumsum(lo**k / k_fact, dim=0)
        )
    raise ValueError(kind)


def multistep_weights(kind: str, u: Tensor, i: int, n: int) -> Tensor:
    r"""``_adams_bashforth`` / ``_exponential_adams_bashforth``: fp64 Vandermonde solve, rounded back
    to the dtype of ``u`` -- azula/sample.py:486-508 (and :652-685, :765-794, :884-912)."""
    t = u.to(torch.float64)
    n = min(n, i + 1)
    k = torch.arange(n)
    V = t[i + 1 - n : i + 1] ** k[:, None]
    return torch.linalg.solve(V, _basis_integrals(kind, t, i, k)).to(u.dtype)


def sample_multistep(mean_fn, x, kind: str, order: int = 2, schedule=vp_schedule, steps: int = 64, **kwargs) -> Tensor:
    r"""zAB (azula/sample.py:519-546), vAB (:593-620), zEAB (:688-715), xEAB (:797-821), REAB (:915-950)."""
    time = timesteps(steps=steps)
    alpha, sigma = schedule(time)
    if kind == "zAB":
        u = sigma / alpha
    elif kind == "vAB":
        u = sigma / (alpha + sigma)
    else:
        u = sigma.log() - alpha.log()
    x_t, buffer = x, []
    for i, t in enumerate(time[:-1].unbind()):
        alpha_t, sigma_t, alpha_s, sigma_s = alpha[i], sigma[i], alpha[i + 1], sigma[i + 1]
        mean = mean_fn(x_t, t, **kwargs)
        if kind in ("zAB", "zEAB"):
            pred = (x_t - alpha_t * mean) / sigma_t
        elif kind == "vAB":
            pred = 1 / sigma_t * x_t - (1 + alpha_t / sigma_t) * mean
        elif kind == "xEAB":
            pred = mean
        else:
            a_t = sigma_t**2 / (alpha_t**2 + sigma_t**2)
            b_t = sigma_t * torch.rsqrt(alpha_t**2 + sigma_t**2)
            pred = (1 - a_t) / b_t / alpha_t * x_t - 1 / b_t * mean
        buffer.append(pred)
        if len(buffer) > order:
            buffer.pop(0)
        coeffs = multistep_weights(kind, u, i, order)
        integral = sum(b * c for b, c in zip(buffer, coeffs, strict=True))
        if kind in ("zAB", "zEAB"):
            x_t = alpha_s / alpha_t * x_t + alpha_s * integral
        elif kind == "vAB":
            x_t = (alpha_s + sigma_s) / (alpha_t + sigma_t) * x_t + (alpha_s + sigma_s) * integral
        elif kind == "xEAB":
            x_t = sigma_s / sigma_t * x_t - sigma_s * integral
        else:
            x_t = (
                torch.sqrt((alpha_s**2 + sigma_s**2) / (alpha_t**2 + sigma_t**2)) * x_t
                + torch.sqrt(alpha_s**2 + sigma_t**2) * integral
            )
    return x_t


def sample_pc(mean_fn, x, schedule=vp_schedule, steps: int = 64, corrections: int = 1, delta: float = 0.01,
              eps_list=None, record_eps=None, **kwargs) -> Tensor:
    r"""PCSampler loop -- azula/sample.py:975-993 (corrector noise drawn after each denoiser call)."""
    x_t, j = x, 0
    for t, s in time_pairs(steps=steps).unbind():
        alpha_s, sigma_s = schedule(s)
        alpha_t, sigma_t = schedule(t)
        for _ in range(corrections):
            mean = mean_fn(x_t, t, **kwargs)
            eps = torch.randn_like(x_t) if eps_list is None else eps_list[j]
            j += 1
            if record_eps is not None:
                record_eps.append(eps)
            x_t = alpha_t * mean + math.sqrt(1 - delta) * (x_t - alpha_t * mean) + math.sqrt(delta) * sigma_t * eps
        mean = mean_fn(x_t, t, **kwargs)
        x_t = alpha_s * mean + sigma_s / sigma_t * (x_t - alpha_t * mean)
    return x_t


# --------------------------------------------------------------------------- JiT plugin denoiser
def rectified_schedule(t: Tensor, alpha_min: float = 1e-3, sigma_min: float = 1e-3):
    r"""alpha_t = t alpha_min + (1 - t), sigma_t = t + (1 - t) sigma_min -- azula/noise.py:186-190."""
    return t * alpha_min + (1 - t), t + (1 - t) * sigma_min


def cosine_schedule(t: Tensor, alpha_min: float = 1e-3, sigma_min: float = 1e-3):
    r"""alpha_t = cos(acos(alpha_min) t), sigma_t = sqrt(1 - alpha_t^2 + sigma_min^2)  -- azula/noise.py:147-154."""
    alpha = torch.cos(math.acos(alpha_min) * t)
    return alpha, torch.sqrt(1 - torch.cos(math.acos(alpha_min) * t) ** 2 + sigma_min**2)


def simple_mean(backbone, x_t: Tensor, t: Tensor, schedule=vp_schedule, backbone_dtype: torch.dtype = torch.float32, **kwargs) -> Tensor:
    r"""SimpleDenoiser: mu = F(c_in x_t, c_time)  -- azula/denoise.py:201-228."""
    alpha_t, sigma_t = schedule(t)
    alpha_t, sigma_t = _expand(alpha_t, x_t.ndim), _expand(sigma_t, x_t.ndim)
    c_in = torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_time = torch.log(sigma_t / alpha_t).reshape_as(t)
    return backbone((c_in * x_t).to(backbone_dtype), c_time.to(backbone_dtype), **kwargs).to(x_t)


def jit_mean(backbone, x_t: Tensor, t: Tensor, label: Tensor | None = None, num_classes: int = 1000,
             schedule=rectified_schedule) -> Tensor:
    r"""JITDenoiser.forward -- plugins/jit/__init__.py:60-102: c_in = 1 / (alpha + sigma),
    c_time = alpha / (alpha + sigma), missing label = the extra "null" class."""
    alpha_t, sigma_t = schedule(t)
    alpha_t, sigma_t = _expand(alpha_t, x_t.ndim), _expand(sigma_t, x_t.ndim)
    c_in = 1 / (alpha_t + sigma_t)
    c_time = (alpha_t / (alpha_t + sigma_t)).flatten()
    if label is None:
        label = torch.as_tensor(num_classes)
    return backbone(c_in * x_t, c_time, label.expand(x_t.shape[0])).to(x_t)
