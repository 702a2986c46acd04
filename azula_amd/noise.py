r"""Noise schedules on the sampling path (counterpart of ``azula.noise``).

A schedule maps a time ``t`` in [0, 1] to the signal scale ``alpha_t`` and the noise scale ``sigma_t``
of the perturbation kernel N(alpha_t x, sigma_t^2 I) (reference ``azula/noise.py:49-63``).

On the hot path schedules are evaluated on the HOST, on 0-d CPU tensors, once per sampling run; the
per-step coefficient table they produce is uploaded in one copy and indexed on the device (see
``azula_amd.sample``).  Evaluated on device tensors they are ordinary torch elementwise ops, as in
the reference.
"""

from __future__ import annotations

import abc
from math import acos, log

import torch
from torch import Tensor

__all__ = ["Schedule", "VPSchedule", "VESchedule", "CosineSchedule", "RectifiedSchedule", "DecaySchedule"]


class Schedule(abc.ABC):
    r"""Interface: ``schedule(t) -> (alpha_t, sigma_t)``, each shaped like ``t``."""

    @abc.abstractmethod
    def __call__(self, t: Tensor) -> tuple[Tensor, Tensor]:
        ...


class VPSchedule(Schedule):
    r"""Variance preserving: alpha_t = exp(t^2 log alpha_min), sigma_t = sqrt(1 - alpha_t^2 + sigma_min^2)
    (reference ``azula/noise.py:99-129``; defaults 1e-3 / 1e-3, the ADM plugin uses 1e-2 / 1e-2)."""

    def __init__(self, alpha_min: float = 1e-3, sigma_min: float = 1e-3) -> None:
        self.alpha_min, self.sigma_min = alpha_min, sigma_min

    def alpha(self, t: Tensor) -> Tensor:
        # same op order as the reference so that host tables are bit-identical to its CPU values
        return torch.exp(log(self.alpha_min) * t**2)

    def sigma(self, t: Tensor) -> Tensor:
        return torch.sqrt(1 - self.alpha(t) ** 2 + self.sigma_min**2)

    def __call__(self, t: Tensor) -> tuple[Tensor, Tensor]:
        return self.alpha(t), self.sigma(t)


class VESchedule(Schedule):
    r"""Variance exploding: alpha_t = 1, sigma_t log-linear between sigma_min and sigma_max
    (reference ``azula/noise.py:66-96``)."""

    def __init__(self, sigma_min: float = 1e-3, sigma_max: float = 1e3) -> None:
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def alpha(self, t: Tensor) -> Tensor:
        return torch.ones_like(t)

    def sigma(self, t: Tensor) -> Tensor:
        return torch.exp((1 - t) * log(self.sigma_min) + t * log(self.sigma_max))

    def __call__(self, t: Tensor) -> tuple[Tensor, Tensor]:
        return self.alpha(t), self.sigma(t)


class CosineSchedule(VPSchedule):
    r"""alpha_t = cos(t arccos alpha_min), sigma_t as the VP schedule (reference ``azula/noise.py:132-157``)."""

    def alpha(self, t: Tensor) -> Tensor:
        return torch.cos(acos(self.alpha_min) * t)


class RectifiedSchedule(Schedule):
    r"""Straight-line (rectified flow / flow matching) schedule: alpha_t = t alpha_min + (1 - t),
    sigma_t = t + (1 - t) sigma_min (reference ``azula/noise.py:160-190``)."""

    def __init__(self, alpha_min: float = 1e-3, sigma_min: float = 1e-3) -> None:
        self.alpha_min, self.sigma_min = alpha_min, sigma_min

    def warp(self, t: Tensor) -> Tensor:
        return t

    def alpha(self, t: Tensor) -> Tensor:
        t = self.warp(t)
        return t * self.alpha_min + (1 - t)

    def sigma(self, t: Tensor) -> Tensor:
        t = self.warp(t)
        return t + (1 - t) * self.sigma_min

    def __call__(self, t: Tensor) -> tuple[Tensor, Tensor]:
        return self.alpha(t), self.sigma(t)


class DecaySchedule(RectifiedSchedule):
    r"""Rectified schedule on the warped time tau = (1 - gamma^t) / (1 - gamma): small gamma spends more of
    [0, 1] at high signal-to-noise ratios (reference ``azula/noise.py:193-231``)."""

    def __init__(self, alpha_min: float = 1e-3, sigma_min: float = 1e-3, gamma: float = 0.1) -> None:
        super().__init__(alpha_min, sigma_min)
        self.gamma = gamma

    def tau(self, t: Tensor) -> Tensor:
        return (1 - self.gamma**t) / (1 - self.gamma)

    warp = tau
