r"""Building-block layers for user-defined backbones (counterpart of ``azula.nn.layers``).

The compiled backbones of this package (``UNet``, ``ViT``, ADM ``UNetModel``) do not use these modules:
their norms, activations and patch re-layouts are fused into HIP kernels.  The classes below exist so
that custom ``nn.Module`` backbones written against ``azula.nn.layers`` (e.g. the reference tests'
``Dummy`` MLP with a ``SineEncoding``) import unchanged; they are plain torch modules and run on
whatever device their inputs live on, exactly like the reference's.
"""

from __future__ import annotations

import math
from collections.abc import Sequence

import torch
import torch.nn as nn
from torch import Tensor

from .utils import promote_dtype

__all__ = ["ConvNd", "LayerNorm", "Patchify", "RMSNorm", "ReLU2", "SineEncoding", "SwiGLU", "Unpatchify"]


def ConvNd(in_channels: int, out_channels: int, spatial: int = 2, identity_init: bool = False, **kwargs) -> nn.Module:
    r"""N-d convolution factory (``Linear`` for spatial = 0), optionally initialised as a near-identity:
    the first ``in_channels`` filters are scaled by 1e-2 and get a unit centre tap (reference
    ``azula/nn/layers.py:25-68``)."""
    kinds = {0: nn.Linear, 1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
    if spatial not in kinds:
        raise NotImplementedError(f"spatial={spatial}")
    conv = kinds[spatial](in_channels, out_channels, **kwargs)
    if identity_init:
        w = conv.weight.data[:in_channels]
        centre = tuple(k // 2 for k in w.shape[2:])
        w.mul_(1e-2)
        for i in range(min(in_channels, out_channels)):
            w[(i, i, *centre)] += 1.0
    return conv


def relu2(x: Tensor) -> Tensor:
    return torch.relu(x).square()


class ReLU2(nn.Module):
    r"""y = max(x, 0)^2 (reference ``azula/nn/layers.py:71-86``)."""

    def forward(self, x: Tensor) -> Tensor:
        return relu2(x)


def swiglu(x: Tensor) -> Tensor:
    pairs = x.unflatten(-1, (-1, 2))
    return pairs[..., 0] * torch.nn.functional.silu(pairs[..., 1])


class SwiGLU(nn.Module):
    r"""(*, 2C) -> (*, C): x1 * silu(x2) over interleaved pairs (reference ``azula/nn/layers.py:89-110``)."""

    def forward(self, x: Tensor) -> Tensor:
        return swiglu(x)


@promote_dtype
def layer_norm(x: Tensor, /, dim: int | Sequence[int] = -1, eps: float = 1e-5) -> Tensor:
    r"""Standardisation along ``dim`` with the UNBIASED variance and no affine (reference
    ``azula/nn/layers.py:152-155``)."""
    var, mean = torch.var_mean(x, dim=dim, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps)


@promote_dtype
def rms_norm(x: Tensor, /, dim: int | Sequence[int] = -1, eps: float = 1e-5) -> Tensor:
    r"""x / sqrt(mean(x^2) + eps) along ``dim`` (reference ``azula/nn/layers.py:193-195``)."""
    return x * torch.rsqrt(x.square().mean(dim=dim, keepdim=True) + eps)


class _DimNorm(nn.Module):
    def __init__(self, dim: int | Sequence[int], eps: float = 1e-5) -> None:
        super().__init__()
        self.dim, self.eps = dim, eps

    def extra_repr(self) -> str:
        return f"dim={self.dim}"


class LayerNorm(_DimNorm):
    def forward(self, x: Tensor) -> Tensor:
        return layer_norm(x, dim=self.dim, eps=self.eps)


class RMSNorm(_DimNorm):
    def forward(self, x: Tensor) -> Tensor:
        return rms_norm(x, dim=self.dim, eps=self.eps)


class _Patch(nn.Module):
    r"""'... Z (A a) (B b) ...' <-> '... (Z a b ...) A B ...' (or channel-last), without einops."""

    def __init__(self, patch_shape: Sequence[int], channel_last: bool, inverse: bool) -> None:
        super().__init__()
        self.patch_shape, self.channel_last, self.inverse = tuple(patch_shape), channel_last, inverse

    def forward(self, x: Tensor) -> Tensor:
        n = len(self.patch_shape)
        p = self.patch_shape
        if not self.inverse:
            lead, Z, sizes = x.shape[: -n - 1], x.shape[-n - 1], x.shape[-n:]
            grid = [s // q for s, q in zip(sizes, p)]
            x = x.reshape(*lead, Z, *[v for g, q in zip(grid, p) for v in (g, q)])
            L = len(lead)
            outer = [L + 1 + 2 * i for i in range(n)]
            inner = [L + 2 + 2 * i for i in range(n)]
            if self.channel_last:
                return x.permute(*range(L), *outer, L, *inner).reshape(*lead, *grid, Z * math.prod(p))
            return x.permute(*range(L), L, *inner, *outer).reshape(*lead, Z * math.prod(p), *grid)
        if self.channel_last:
            lead, grid, F_ = x.shape[: -n - 1], x.shape[-n - 1 : -1], x.shape[-1]
            Z = F_ // math.prod(p)
            x = x.reshape(*lead, *grid, Z, *p)
            L = len(lead)
            order = [L + n] + [v for i in range(n) for v in (L + i, L + n + 1 + i)]
        else:
            lead, F_, grid = x.shape[: -n - 1], x.shape[-n - 1], x.shape[-n:]
            Z = F_ // math.prod(p)
            x = x.reshape(*lead, Z, *p, *grid)
            L = len(lead)
            order = [L] + [v for i in range(n) for v in (L + 1 + n + i, L + 1 + i)]
        return x.permute(*range(L), *order).reshape(*lead, Z, *[g * q for g, q in zip(grid, p)])


def Patchify(patch_shape: Sequence[int], channel_last: bool = False) -> nn.Module:
    r"""Patch-to-channel layer (reference ``azula/nn/layers.py:198-222``)."""
    return _Patch(patch_shape, channel_last, inverse=False)


def Unpatchify(patch_shape: Sequence[int], channel_last: bool = False) -> nn.Module:
    r"""Channel-to-patch layer (reference ``azula/nn/layers.py:225-247``)."""
    return _Patch(patch_shape, channel_last, inverse=True)


@promote_dtype
def sine_encoding(x: Tensor, /, features: int, omega: float = 1e4) -> Tensor:
    r"""(*,) -> (*, D): sin(x w^(-2i/D)) block followed by the cos block (reference
    ``azula/nn/layers.py:286-299``)."""
    freqs = torch.exp(math.log(1 / omega) * torch.linspace(0, 1, features // 2, dtype=x.dtype, device=x.device))
    arg = x.unsqueeze(dim=-1) * freqs
    return torch.cat((torch.sin(arg), torch.cos(arg)), dim=-1)


class SineEncoding(nn.Module):
    r"""Sinusoidal positional encoding with ``features`` (even) outputs and maximum frequency ``omega``."""

    def __init__(self, features: int, omega: float = 1e4) -> None:
        super().__init__()
        assert features % 2 == 0
        self.features, self.omega = features, omega

    def forward(self, x: Tensor) -> Tensor:
        return sine_encoding(x, features=self.features, omega=self.omega)
