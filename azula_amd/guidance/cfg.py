r"""Classifier-free guidance -- drop-in for ``azula.guidance.cfg`` (reference ``cfg.py:19-65``).

mu = mu(x_t | c+) + g * (mu(x_t | c+) - mu(x_t | c-)).  The reference makes two sequential,
un-batched denoiser calls and three elementwise passes; inside a fused sampler the two backbone
programs run back to back in the step graph and the combine is folded into the transition kernel
(``az_transition_f32`` with ``F_neg``), so guidance adds no pass over the latent.
"""

from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from .. import _lib
from ..denoise import Denoiser, DiracPosterior
from ..noise import Schedule

__all__ = ["CFGDenoiser"]


class CFGDenoiser(Denoiser):
    r"""Creates a CFG denoiser module around ``denoiser``."""

    def __init__(self, denoiser: Denoiser) -> None:
        super().__init__()
        self.denoiser = denoiser

    @property
    def schedule(self) -> Schedule:
        return self.denoiser.schedule

    @torch.no_grad()
    @_lib.on_device
    def forward(
        self,
        x_t: Tensor,
        t: Tensor,
        positive: dict[str, Any],
        negative: dict[str, Any] = {},  # noqa: B006
        guidance: float | Tensor = 1.0,
        **kwargs,
    ) -> DiracPosterior:
        q_pos = self.denoiser(x_t, t, **positive, **kwargs)
        q_neg = self.denoiser(x_t, t, **negative, **kwargs)
        if not x_t.is_cuda:  # host tensors: reference op sequence
            return DiracPosterior(mean=q_pos.mean + guidance * (q_pos.mean - q_neg.mean))
        if q_pos.mean.dtype == torch.float64:  # fp64 means (Sampler(dtype=float64)): pos + g * (pos - neg) in fp64
            from ..denoise import axpby_wide

            one = torch.ones((), dtype=torch.float64)
            diff = axpby_wide(one, q_pos.mean, -one, q_neg.mean)
            g64 = torch.as_tensor(guidance, dtype=torch.float64).reshape(())
            return DiracPosterior(mean=axpby_wide(one, q_pos.mean, g64, diff))
        pos, neg = q_pos.mean.contiguous(), q_neg.mean.contiguous()
        g = torch.as_tensor(guidance, dtype=torch.float32, device=x_t.device).reshape(1)
        mean = torch.empty_like(pos)
        _lib.call("az_cfg_combine_f32", mean.data_ptr(), pos.data_ptr(), neg.data_ptr(), g.data_ptr(), pos.numel(), _lib.stream_ptr())
        return DiracPosterior(mean=mean)

    # -- fused sampling ---------------------------------------------------------------------------
    def _az_fused(self, x: Tensor, kwargs: dict, cur_coef: Tensor):
        from ..sample import FusedDenoiser

        inner = self.denoiser
        make = getattr(inner, "_az_programs", None)
        if make is None or set(kwargs) - {"positive", "negative", "guidance"} or "positive" not in kwargs:
            return None
        guidance = kwargs.get("guidance", 1.0)
        if torch.is_tensor(guidance):
            if guidance.numel() != 1:
                return None
            guidance = float(guidance)
        pos, neg = dict(kwargs["positive"]), dict(kwargs.get("negative", {}))
        programs = make(x, [pos, neg], cur_coef)
        if programs is None:
            return None
        # each program's prepare() looks up its own label set
        for i, p in enumerate(programs):
            orig = p.prepare

            def prepare(call_kwargs: dict, orig=orig, i=i) -> None:
                labels = {0: call_kwargs["positive"].get("label"), 1: call_kwargs.get("negative", {}).get("label")}
                orig({"_az_labels": labels})

            p.prepare = prepare
        return FusedDenoiser(
            coefficients=inner.host_coefficients, programs=programs, guidance=float(guidance), clip=inner._clip()
        )
