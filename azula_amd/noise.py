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
from math import log

import torch
from torch import Tensor

__all__ = ["Schedule", "VPSchedule", "VESchedule"]


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
