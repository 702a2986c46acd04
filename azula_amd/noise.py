r"""Noise schedules -- drop-in for ``azula.noise`` on the sampling path.

A schedule maps a time ``t`` to ``(alpha_t, sigma_t)`` (reference ``azula/noise.py:49-63``).
On the hot path schedules are evaluated on the HOST, on 0-d CPU tensors, in the reference's
op order, once per sampling run; the resulting per-step coefficient table is uploaded in one
copy and indexed on the device (see ``azula_amd.sample``).  Called on device tensors the
classes behave exactly like the reference's (plain torch elementwise ops).
"""

from __future__ import annotations

import abc
import math

import torch
from torch import Tensor

__all__ = ["Schedule", "VPSchedule", "VESchedule"]


class Schedule(abc.ABC):
    r"""Abstract noise schedule (reference ``azula/noise.py:49-63``)."""

    @abc.abstractmethod
    def __call__(self, t: Tensor) -> tuple[Tensor, Tensor]:
        r"""t: (*) -> (alpha_t, sigma_t), each (*)."""


class VPSchedule(Schedule):
    r"""Variance preserving schedule (reference ``azula/noise.py:99-129``).

    alpha_t = exp(t^2 log alpha_min),  sigma_t = sqrt(1 - alpha_t^2 + sigma_min^2).
    """

    def __init__(self, alpha_min: float = 1e-3, sigma_min: float = 1e-3) -> None:
        self.alpha_min = alpha_min
        self.sigma_min = sigma_min

    def __call__(self, t: Tensor) -> tuple[Tensor, Tensor]:
        return self.alpha(t), self.sigma(t)

    def alpha(self, t: Tensor) -> Tensor:
        return torch.exp(math.log(self.alpha_min) * t**2)

    def sigma(self, t: Tensor) -> Tensor:
        return torch.sqrt(1 - self.alpha(t) ** 2 + self.sigma_min**2)


class VESchedule(Schedule):
    r"""Variance exploding schedule (reference ``azula/noise.py:66-96``): alpha = 1,
    sigma_t = exp((1 - t) log sigma_min + t log sigma_max)."""

    def __init__(self, sigma_min: float = 1e-3, sigma_max: float = 1e3) -> None:
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max

    def __call__(self, t: Tensor) -> tuple[Tensor, Tensor]:
        return self.alpha(t), self.sigma(t)

    def alpha(self, t: Tensor) -> Tensor:
        return torch.ones_like(t)

    def sigma(self, t: Tensor) -> Tensor:
        return torch.exp((1 - t) * math.log(self.sigma_min) + t * math.log(self.sigma_max))
