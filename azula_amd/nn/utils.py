r"""Module helpers (reference ``azula/nn/utils.py:24-42,172-188``)."""

from __future__ import annotations

from functools import reduce, wraps

import torch

__all__ = ["get_module_dtype", "promote_dtype", "skip_init"]


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
