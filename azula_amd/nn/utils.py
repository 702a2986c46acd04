r"""Module helpers (reference ``azula/nn/utils.py:24-42,172-188``)."""

from __future__ import annotations

from functools import reduce, wraps

import torch

__all__ = ["get_module_dtype", "get_module_device", "promote_dtype", "skip_init", "checkpoint", "backbone_io_dtype", "HALF_DTYPES"]


def get_module_dtype(module: torch.nn.Module) -> torch.dtype | None:
    r"""First floating-point dtype among the module's parameters, then buffers
    (reference ``azula/nn/utils.py:24-42``)."""
    for p in module.parameters():
        if torch.is_floating_point(p):
            return p.dtype
    for b in module.buffers():
        if torch.is_floating_point(b):
            return b.dtype
    return None


class skip_init(torch.overrides.TorchFunctionMode):
    r"""Context in which ``torch.nn.init.*`` calls are no-ops (reference
    ``azula/nn/utils.py:172-188``) -- used when weights are about to be loaded."""

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if getattr(func, "__module__", None) == "torch.nn.init":
            return kwargs["tensor"] if "tensor" in kwargs else args[0]
        return func(*args, **kwargs)


def promote_dtype(f, min_dtype: torch.dtype = torch.float32):
    r"""Decorator: positional tensor arguments are promoted to at least ``min_dtype`` for the call
    and the outputs cast back to the inputs' common dtype (reference ``azula/nn/utils.py:191-221``;
    the MPS special case is irrelevant on ROCm)."""

    @wraps(f)
    def g(*args, **kwargs):
        common = reduce(torch.promote_types, (a.dtype for a in args))
        outs = f(*(a.to(torch.promote_types(a.dtype, min_dtype)) for a in args), **kwargs)
        if torch.is_tensor(outs):
            return outs.to(dtype=common)
        return tuple(o.to(dtype=common) for o in outs)

    return g


HALF_DTYPES = (torch.float16, torch.bfloat16)


def backbone_io_dtype(module: torch.nn.Module, x: torch.Tensor, who: str) -> torch.dtype:
    r"""Validates the (module, input) pair of a compiled backbone and returns the dtype its output must have.

    fp32 modules take fp32 inputs.  A module cast with ``.half()`` / ``.bfloat16()`` (the reference's mixed-precision
    route: the denoiser casts ``c_in x_t`` to the backbone's dtype and the output back, ``azula/denoise.py:314-320``)
    is accepted too: every convolution / token GEMM of its plan then runs on ``v_mfma_f32_32x32x16_{bf16,f16}`` with
    weights packed in the module's dtype, fp32 accumulation and fp32 activations (``az_conv2d_{bf16,f16}_f32``);
    norms, attention and the small modulation linears stay fp32 on up-converted parameters.  The result is within the
    reference's own fp16-vs-fp32 tolerance (``tests/test_nn_unet.py:78-91``) of the fp32 output."""
    if not x.is_cuda:
        raise RuntimeError(
            f"{who} executes only on an AMD GPU (gfx950 HIP kernels); there is no CPU fallback. "
            "Move the module and its inputs to 'cuda'."
        )
    p = next(module.parameters())
    ok = (p.dtype == torch.float32 and x.dtype == torch.float32) or (p.dtype in HALF_DTYPES and x.dtype in (p.dtype, torch.float32))
    if p.device != x.device or not ok:
        raise RuntimeError(
            f"{who} needs parameters and inputs on the same GPU, fp32 / fp32 or half / (half | fp32); "
            f"got {p.dtype} on {p.device} and {x.dtype} on {x.device}"
        )
    return x.dtype


def get_module_device(module: torch.nn.Module) -> torch.device | None:
    r"""Device of the first materialised parameter (then buffer) of ``module``, ``None`` if it has none
    (reference ``azula/nn/utils.py:45-68``; an accelerate hook's ``execution_device`` wins when present)."""
    for m in module.modules():
        hook = getattr(m, "_hf_hook", None)
        if getattr(hook, "execution_device", None) is not None:
            return hook.execution_device
        for t in (*m.parameters(recurse=False), *m.buffers(recurse=False)):
            if t.device.type != "meta":
                return t.device
    return None


def checkpoint(f, reentrant: bool = False):
    r"""Activation checkpointing wrapper (reference ``azula/nn/utils.py:123-188``).  A training-time memory tool: on
    the sampling path gradients are disabled and the wrapped function is simply called; with gradients enabled it
    defers to ``torch.utils.checkpoint`` (``reentrant`` selects its mode)."""

    @wraps(f)
    def g(*args, **kwargs):
        if not torch.is_grad_enabled():
            return f(*args, **kwargs)
        import torch.utils.checkpoint as tuc

        return tuc.checkpoint(f, *args, use_reentrant=reentrant, **kwargs)

    return g
