r"""Ablated diffusion model (ADM) plugin -- drop-in for ``azula.plugins.adm``.

    from azula_amd.plugins import adm
    denoiser = adm.load_model("imagenet_256x256").to("cuda")     # needs the checkpoint in the hub cache
    denoiser = adm.make_model(**adm.load_cards(adm)["imagenet_256x256"].config)   # random init

Reference: ``azula/plugins/adm/__init__.py:33-202`` (Dhariwal & Nichol, 2021).
"""

from __future__ import annotations

import os

import ctypes as C
import math
from collections.abc import Sequence

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib
from ...denoise import Denoiser, GaussianPosterior, _expand_like, require_f32_cuda
from ...engine import Tape, transition_args
from ...hub import download
from ...nn.utils import get_module_dtype, skip_init
from ...noise import Schedule, VPSchedule
from ..utils import load_cards
from . import unet

__all__ = ["AblatedDenoiser", "load_model", "make_model", "load_cards"]


def discrete_sigmas(discrete_schedule: str, discrete_steps: int) -> Tensor:
    r"""sigma table of the discrete training schedule, fp64 -> default dtype
    (reference ``azula/plugins/adm/__init__.py:66-84``)."""
    if discrete_schedule == "linear":
        beta = torch.linspace(0.1 / discrete_steps, 20.0 / discrete_steps, discrete_steps, dtype=torch.float64)
    elif discrete_schedule == "cosine":
        t = torch.linspace(0, 1, discrete_steps + 1, dtype=torch.float64)
        alpha_bar = torch.cos((t + 0.008) / 1.008 * torch.pi / 2) ** 2
        beta = 1 - alpha_bar[1:] / alpha_bar[:-1]
        beta = torch.clip(beta, max=0.999)
    else:
        raise ValueError(f"Unknown discrete schedule '{discrete_schedule}'.")
    alpha_bar = torch.cumprod(1 - beta, dim=0)
    return torch.sqrt(1 - alpha_bar).to(torch.get_default_dtype())


def adm_coefficients(alpha_t: Tensor, sigma_t: Tensor, sigmas: Tensor):
    r"""c_in, c_out, c_skip, time index, c_var in the reference's op order (``__init__.py:109-114``)."""
    c_in = torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_out = -sigma_t / alpha_t
    c_skip = 1 / alpha_t
    c_time = sigma_t * torch.rsqrt(alpha_t**2 + sigma_t**2)
    c_time = torch.searchsorted(sigmas, c_time.flatten())
    c_var = sigma_t**2 / (alpha_t**2 + sigma_t**2)
    return c_in, c_out, c_skip, c_time, c_var


class AblatedDenoiser(Denoiser):
    r"""Epsilon-prediction denoiser around a guided-diffusion UNet (reference
    ``azula/plugins/adm/__init__.py:33-136``): ``mean = (x_t - sigma_t eps) / alpha_t``,
    optionally clipped to [-1, 1] in eval mode, ``var = c_var * exp(log_var)`` if learned."""

    def __init__(
        self,
        backbone: nn.Module,
        schedule: Schedule | None = None,
        clip_mean: bool = False,
        learn_var: bool = False,
        discrete_schedule: str = "linear",
        discrete_steps: int = 1000,
    ) -> None:
        super().__init__()
        self.backbone = backbone
        self.schedule = VPSchedule(alpha_min=1e-2, sigma_min=1e-2) if schedule is None else schedule
        self.clip_mean = clip_mean
        self.learn_var = learn_var
        self.register_buffer("sigmas", discrete_sigmas(discrete_schedule, discrete_steps))

    def _clip(self) -> tuple[float, float]:
        return (-1.0, 1.0) if (not self.training and self.clip_mean) else (-math.inf, math.inf)

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x_t: Tensor, t: Tensor, label: Tensor | None = None, **kwargs) -> GaussianPosterior:
        alpha_t, sigma_t = self.schedule(t)
        alpha_t, sigma_t = _expand_like(alpha_t, x_t.ndim), _expand_like(sigma_t, x_t.ndim)
        sig = self.sigmas if self.sigmas.device == alpha_t.device else self.sigmas.to(alpha_t.device)
        c_in, c_out, c_skip, c_time, c_var = adm_coefficients(alpha_t, sigma_t, sig)
        if not x_t.is_cuda:
            raise RuntimeError("azula_amd ADM denoisers execute only on an AMD GPU (no CPU fallback)")
        from ...denoise import axpby_wide, is_wide, precondition, precondition_wide

        if is_wide(x_t, alpha_t):  # fp64 time grid / latents: fp64 elementwise path around the fp32 backbone
            dev = x_t.device
            x_in = precondition_wide(x_t, c_in)
            output = self.backbone(x_in, c_time.to(dev), y=label, **kwargs)
            Cc = x_t.shape[1]
            eps_hat = output[:, :Cc].to(x_t.dtype).contiguous() if self.learn_var else output.to(x_t.dtype)
            mean = axpby_wide(c_skip, x_t, c_out, eps_hat)
            lo, hi = self._clip()
            if lo > -math.inf:
                mean = torch.clip(mean, min=lo, max=hi)
            var = c_var.to(dev) * torch.exp(output[:, Cc:].to(x_t.dtype)) if self.learn_var else c_var.to(dev)
            return GaussianPosterior(mean=mean, var=var)
        require_f32_cuda(x_t, "AblatedDenoiser")

        dev = x_t.device
        x_t = x_t.contiguous()
        x_in = precondition(x_t, c_in.to(dev))
        output = self.backbone(x_in, c_time.to(dev), y=label, **kwargs).contiguous()
        B, Cc = x_t.shape[0], x_t.shape[1]
        inner = x_t.numel() // (B * Cc)
        # mean = clip(c_skip * x_t + c_out * eps[:, :C]) through the transition kernel's mean output
        lo, hi = self._clip()
        mean = torch.empty_like(x_t)
        if c_skip.numel() == 1:
            row = torch.zeros(_lib.COEF_WORDS, dtype=torch.float32, device=dev)
            col = {n: i for i, n in enumerate(_lib.COEF_FIELDS)}
            row[col["c_skip"]], row[col["c_out"]] = c_skip.reshape(()).to(dev), c_out.reshape(()).to(dev)
            row[col["clip_lo"]], row[col["clip_hi"]] = lo, hi
            scratch = torch.empty_like(x_t)
            a = transition_args(
                x_t=x_t.data_ptr(), F=output.data_ptr(), x_s=scratch.data_ptr(), mean_out=mean.data_ptr(), batch=B,
                channels=Cc, inner=inner, f_channels=output.shape[1], coef=row.data_ptr(),
            )
            _lib.call("az_transition_f32", C.byref(a), _lib.stream_ptr())
        else:  # per-sample times t of shape (B,)
            from ...denoise import postcondition

            mean = postcondition(x_t, output[:, :Cc].contiguous(), c_skip.to(dev), c_out.to(dev))
            if lo > -math.inf:
                mean = torch.clip(mean, min=lo, max=hi)
        if self.learn_var:
            log_var = output[:, Cc:]
            var = c_var.to(dev) * torch.exp(log_var)  # not on the sampling path (samplers read .mean only)
        else:
            var = c_var.to(dev)
        return GaussianPosterior(mean=mean, var=var)

    # -- fused sampling -------------------------------------------------------------------------------
    def host_coefficients(self, alpha_t: Tensor, sigma_t: Tensor) -> dict:
        c_in, c_out, c_skip, idx, _ = adm_coefficients(alpha_t, sigma_t, self.sigmas.detach().cpu())
        return {"c_in": c_in, "c_out": c_out, "c_skip": c_skip, "time_index": idx.reshape(())}

    def _az_programs(self, x: Tensor, kwargs_list: list[dict], cur_coef: Tensor):
        r"""One compiled backbone program per kwargs dict (CFG: positive, negative), all reading
        the same pre-scaled NHWC input buffer."""
        from ...sample import BackboneProgram
        from ...nn.unet import _copy_tape

        bb = self.backbone
        if not isinstance(bb, unet.UNetModel) or x.ndim != bb.dims + 2 or get_module_dtype(bb) not in (torch.float32, torch.float16, torch.bfloat16):
            return None
        B, (H, W) = x.shape[0], _map_size(x)
        D = x.shape[2] if x.ndim == 5 else 1
        if len(kwargs_list) == 2 and bb.num_classes is not None and os.environ.get("AZ_CFG_BATCHED", "1") != "0":
            batched = self._az_cfg_batched(x, kwargs_list, cur_coef)
            if batched is not None:
                return batched
        programs, x_in = [], None
        for i, kw in enumerate(kwargs_list):
            if set(kw) - {"label"}:
                return None
            label = kw.get("label")
            if (label is not None) != (bb.num_classes is not None):
                return None
            rows = B if bb.num_classes is not None else 1
            plan = bb.plan(B, H, W, rows, x.device, x_in=x_in, coef_ptr=cur_coef.data_ptr(), tag=i, D=D)
            x_in = plan.x_in

            def prepare(call_kwargs: dict, plan=plan, key=i) -> None:
                lab = call_kwargs.get("_az_labels", {}).get(key, call_kwargs.get("label"))
                if plan.labels is not None:
                    plan.labels.copy_(lab.to(torch.int64))

            prog = BackboneProgram(
                tape=_copy_tape(plan.tape), x_in=plan.x_in.buf, x_in_cs=plan.x_in.cs, out=plan.out,
                f_channels=bb.out_channels, f_nhwc=False, prepare=prepare,
            )
            prog.tape.keep.append(plan)
            programs.append(prog)
        return programs

    def _az_cfg_batched(self, x: Tensor, kwargs_list: list[dict], cur_coef: Tensor):
        r"""Classifier-free guidance as ONE backbone evaluation on a 2B batch: samples [0, B) carry the
        positive labels, [B, 2B) the negative ones; the transition kernel fills the first half of the input, one copy
        duplicates it, and the two halves of the output are its ``F`` / ``F_neg``.  Same arithmetic per sample as two
        sequential evaluations (the reference's ``cfg.py:60-61``), but the small feature maps fill the chip better."""
        from ...engine import Tape
        from ...nn.unet import _copy_tape
        from ...sample import BackboneProgram

        bb = self.backbone
        if any(set(kw) - {"label"} or kw.get("label") is None for kw in kwargs_list):
            return None
        B, (H, W) = x.shape[0], _map_size(x)
        D = x.shape[2] if x.ndim == 5 else 1
        plan = bb.plan(2 * B, H, W, 2 * B, x.device, coef_ptr=cur_coef.data_ptr(), tag="cfg2b", D=D)
        half_in = B * D * H * W * (plan.x_in.cs if plan.x_in.cs > 0 else plan.x_in.C)  # (channel stride 0: the planar latent layout)
        one = torch.ones(1, dtype=torch.float32, device=x.device)
        tape = Tape()
        tape.add("az_scale_f32", plan.x_in.ptr + 4 * half_in, plan.x_in.ptr, one.data_ptr(), half_in, keep=[one, plan])
        tape.extend(_copy_tape(plan.tape))

        def prepare(call_kwargs: dict, key: int) -> None:
            lab = call_kwargs.get("_az_labels", {}).get(key)
            plan.labels[key * B : (key + 1) * B].copy_(lab.to(torch.int64))

        common = dict(x_in=plan.x_in.buf, x_in_cs=plan.x_in.cs, f_channels=bb.out_channels, f_nhwc=False)
        return [
            BackboneProgram(tape=tape, out=plan.out[:B], prepare=lambda kw: prepare(kw, 0), **common),
            BackboneProgram(tape=Tape(), out=plan.out[B:], prepare=lambda kw: prepare(kw, 1), **common),
        ]

    def _az_fused(self, x: Tensor, kwargs: dict, cur_coef: Tensor):
        from ...sample import FusedDenoiser

        programs = self._az_programs(x, [kwargs], cur_coef)
        if programs is None:
            return None
        return FusedDenoiser(coefficients=self.host_coefficients, programs=programs, clip=self._clip())


def _map_size(x: Tensor) -> tuple[int, int]:
    r"""(H, W) of the feature map a latent is run as: a (B, C, L) signal (``dims=1``) is a one-row image."""
    return (1, x.shape[2]) if x.ndim == 3 else (x.shape[-2], x.shape[-1])


def load_model(name: str, **kwargs) -> Denoiser:
    r"""Loads a pre-trained ADM denoiser from the hub cache (reference ``__init__.py:139-161``)."""
    kwargs.setdefault("map_location", "cpu")
    kwargs.setdefault("weights_only", True)
    card = load_cards(__name__)[name]
    state = torch.load(download(card.url, hash_prefix=card.hash), **kwargs)
    with skip_init():
        denoiser = make_model(**card.config)
    denoiser.backbone.load_state_dict(state)
    return denoiser.eval()


def make_model(
    # Denoiser
    clip_mean: bool = True,
    learn_var: bool = True,
    # Discrete schedule
    discrete_schedule: str = "linear",
    discrete_steps: int = 1000,
    # Data
    image_channels: int = 3,
    image_size: int = 64,
    # Backbone
    attention_resolutions: Sequence[int] = (32, 16, 8),
    channel_mult: Sequence[int] = (1, 2, 3, 4),
    num_channels: int = 128,
    num_classes: int | None = None,
    **kwargs,
) -> Denoiser:
    r"""Initialises an ADM denoiser (reference ``__init__.py:164-202``)."""
    attention_resolutions = {image_size // r for r in attention_resolutions}
    backbone = unet.UNetModel(
        image_size=image_size,
        in_channels=image_channels,
        out_channels=2 * image_channels if learn_var else image_channels,
        model_channels=num_channels,
        channel_mult=channel_mult,
        num_classes=num_classes,
        attention_resolutions=attention_resolutions,
        **kwargs,
    )
    return AblatedDenoiser(
        backbone, clip_mean=clip_mean, learn_var=learn_var, discrete_schedule=discrete_schedule,
        discrete_steps=discrete_steps,
    )
