r"""Module helpers (reference ``azula/nn/utils.py:24-42,172-188``)."""

from __future__ import annotations

import torch

__all__ = ["get_module_dtype", "skip_init"]


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
