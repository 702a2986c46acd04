r"""Just image Transformer (JiT) plugin -- drop-in for ``azula.plugins.jit``.

    from azula_amd.plugins import jit
    denoiser = jit.make_model("JiT-B/16").cuda()

``JITDenoiser`` predicts the clean image directly (reference ``azula/plugins/jit/__init__.py:32-102``):

    mu(x_t | c) = F(c_in x_t, c_time, y = c),   c_in = 1 / (alpha_t + sigma_t),  c_time = alpha_t / (alpha_t + sigma_t)

on the rectified (flow-matching) schedule; a missing label selects the extra "null" class, which is what
``CFGDenoiser``'s negative branch uses.  In a fused sampler this is the transition kernel's Karras form with
c_skip = 0, c_out = 1; the backbone is :class:`JiT` compiled onto the token kernels.
"""

from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib
from ...denoise import Denoiser, DiracPosterior, _expand_like, precondition, require_f32_cuda
from ...hub import download
from ...nn.utils import get_module_dtype, skip_init
from ...noise import RectifiedSchedule, Schedule
from ..utils import load_cards
from .model import JiT, JiT_models

__all__ = ["JITDenoiser", "JiT", "load_model", "make_model"]


def jit_coefficients(alpha_t: Tensor, sigma_t: Tensor):
    r"""(c_in, c_time) in the reference's op order (``plugins/jit/__init__.py:82-83``)."""
    return 1 / (alpha_t + sigma_t), alpha_t / (alpha_t + sigma_t)


class JITDenoiser(Denoiser):
    r"""JiT denoiser.  ``schedule=None`` selects :class:`azula_amd.noise.RectifiedSchedule`."""

    def __init__(self, backbone: nn.Module, schedule: Schedule | None = None, num_classes: int = 1000) -> None:
        super().__init__()
        self.backbone = backbone
        self.schedule = RectifiedSchedule() if schedule is None else schedule
        self.num_classes = num_classes

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x_t: Tensor, t: Tensor, label: Tensor | None = None, **kwargs) -> DiracPosterior:
        alpha_t, sigma_t = self.schedule(t)
        alpha_t, sigma_t = _expand_like(alpha_t, x_t.ndim), _expand_like(sigma_t, x_t.ndim)
        c_in, c_time = jit_coefficients(alpha_t, sigma_t)
        c_time = c_time.flatten()
        B = x_t.shape[0]
        dtype = get_module_dtype(self.backbone) or x_t.dtype
        if label is None:
            label = torch.as_tensor(self.num_classes, device=x_t.device)
        if x_t.is_cuda:
            require_f32_cuda(x_t, "JITDenoiser")
            x_in = precondition(x_t.contiguous(), c_in.to(x_t.device))
        else:
            x_in = c_in * x_t
        output = self.backbone(x_in.to(dtype), c_time.to(dtype), y=label.expand(B), **kwargs).to(x_t)
        return DiracPosterior(mean=output)

    # -- fused sampling -------------------------------------------------------------------------
    def host_coefficients(self, alpha_t: Tensor, sigma_t: Tensor) -> dict:
        c_in, c_time = jit_coefficients(alpha_t, sigma_t)
        return {"c_in": c_in, "c_out": torch.ones_like(c_in), "c_skip": torch.zeros_like(c_in), "c_time": c_time}

    def _clip(self) -> tuple[float, float]:
        return (-math.inf, math.inf)

    def _az_programs(self, x: Tensor, kwargs_list: list[dict], cur_coef: Tensor):
        r"""One compiled backbone program per kwargs dict (CFG: conditional, null class).  Each program owns
        its plan (labels differ) and copies the shared pre-scaled input produced by the transition kernel."""
        from ...engine import Tape
        from ...nn.unet import _copy_tape
        from ...sample import BackboneProgram

        bb = self.backbone
        if not isinstance(bb, JiT) or x.ndim != 4 or get_module_dtype(bb) not in (torch.float32, torch.float16, torch.bfloat16):
            return None
        B = x.shape[0]
        if len(kwargs_list) == 2 and os.environ.get("AZ_CFG_BATCHED", "1") != "0" and not any(set(kw) - {"label"} for kw in kwargs_list):
            # classifier-free guidance as ONE evaluation on a 2B batch: [0, B) positive labels, [B, 2B) negative / null
            from .model import JiTPlan

            plan = JiTPlan(bb, 2 * B, True, x.device)
            half = plan.x_nchw[:B].numel()
            tape = Tape()
            tape.add("az_coef_c_time_f32", plan.t.data_ptr(), cur_coef.data_ptr())
            tape.add("az_scale_f32", plan.x_nchw.data_ptr() + 4 * half, plan.x_nchw.data_ptr(), _one(x.device).data_ptr(), half)
            tape.extend(_copy_tape(plan.tape))
            tape.keep.append(plan)

            def prepare2(call_kwargs: dict, key: int) -> None:
                lab = call_kwargs.get("_az_labels", {}).get(key)
                if lab is None:
                    lab = torch.as_tensor(self.num_classes)
                plan.labels[key * B : (key + 1) * B].copy_(lab.to(device=plan.labels.device, dtype=torch.int64).expand(B))

            common = dict(x_in=plan.x_nchw[:B], x_in_cs=0, f_channels=bb.out_channels, f_nhwc=False)
            return [
                BackboneProgram(tape=tape, out=plan.out[:B], prepare=lambda kw: prepare2(kw, 0), **common),
                BackboneProgram(tape=Tape(), out=plan.out[B:], prepare=lambda kw: prepare2(kw, 1), **common),
            ]
        programs, x_in = [], None
        for i, kw in enumerate(kwargs_list):
            if set(kw) - {"label"}:
                return None
            from .model import JiTPlan

            plan = JiTPlan(bb, B, True, x.device)
            tape = Tape()
            tape.add("az_coef_c_time_f32", plan.t.data_ptr(), cur_coef.data_ptr())
            if x_in is None:
                x_in = plan.x_nchw
            else:  # later programs read the first one's input buffer
                tape.add("az_scale_f32", plan.x_nchw.data_ptr(), x_in.data_ptr(), _one(x.device).data_ptr(), x_in.numel())
            tape.extend(_copy_tape(plan.tape))
            tape.keep.append(plan)

            def prepare(call_kwargs: dict, plan=plan, key=i) -> None:
                lab = call_kwargs.get("_az_labels", {}).get(key, call_kwargs.get("label"))
                if lab is None:
                    lab = torch.as_tensor(self.num_classes)
                plan.labels.copy_(lab.to(device=plan.labels.device, dtype=torch.int64).expand(B))

            programs.append(BackboneProgram(
                tape=tape, x_in=x_in, x_in_cs=0, out=plan.out, f_channels=bb.out_channels, f_nhwc=False, prepare=prepare,
            ))
        return programs

    def _az_fused(self, x: Tensor, kwargs: dict, cur_coef: Tensor):
        from ...sample import FusedDenoiser

        programs = self._az_programs(x, [kwargs], cur_coef)
        if programs is None:
            return None
        return FusedDenoiser(coefficients=self.host_coefficients, programs=programs)


_ONES: dict = {}


def _one(device) -> Tensor:
    key = str(device)
    if key not in _ONES:
        _ONES[key] = torch.ones(1, dtype=torch.float32, device=device)
    return _ONES[key]


def load_model(name: str, ema: bool = True, **kwargs) -> Denoiser:
    r"""Loads a pre-trained JiT denoiser from the hub cache (reference ``plugins/jit/__init__.py:105-145``;
    this build never downloads -- ``azula_amd.hub.download`` resolves an existing cache entry or raises)."""
    kwargs.setdefault("map_location", "cpu")
    kwargs.setdefault("weights_only", True)
    card = load_cards(__name__)[name]
    folder = download(card.url, hash_prefix=card.hash, extract=True)
    state = torch.load(os.path.join(folder, "checkpoint-last.pth"), **kwargs)
    state = state["model_ema1" if ema else "model"]
    state = {k.removeprefix("net."): v for k, v in state.items()}
    with skip_init():
        denoiser = make_model(**card.config)
    denoiser.backbone.load_state_dict(state)
    return denoiser.eval()


def make_model(model: str = "JiT-B/16", **kwargs) -> Denoiser:
    r"""Initialises a JiT denoiser (reference ``plugins/jit/__init__.py:148-156``)."""
    backbone = JiT_models[model](**kwargs)
    return JITDenoiser(backbone, num_classes=backbone.num_classes)
