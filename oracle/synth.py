r"""Deterministic synthetic weights for parity tests -- TEST INFRASTRUCTURE ONLY.

Golden fixtures do not store network weights (they would be MBs); they store the *recipe*:
``synth_state_dict(shapes, seed)`` regenerates the same tensors from the parameter shapes
alone, on any machine with the same torch build, independent of the reference's module
initialisers.  Every tensor is non-degenerate (no zero-init layers, gates of order one), so
parity is never vacuous (SURVEY.md section 7 "Zero-init layers").
"""

from __future__ import annotations

import math
import torch


def synth_tensor(name: str, shape, gen: torch.Generator) -> torch.Tensor:
    shape = tuple(shape)
    if len(shape) > 1:  # weights: N(0, 1 / fan_in)
        fan_in = math.prod(shape[1:])
        if name.endswith("ada_zero"):  # raw (3, C, 1, 1) modulation parameter
            return 0.5 * torch.randn(shape, generator=gen)
        if name.endswith("label_emb.weight"):
            return 0.5 * torch.randn(shape, generator=gen)
        if name.endswith(("pos_embed", "in_context_posemb", "embedding_table.weight")):  # JiT tables: order one
            return 0.5 * torch.randn(shape, generator=gen)
        return torch.randn(shape, generator=gen) / math.sqrt(fan_in)
    if name.endswith(".weight"):  # 1-D "weight" = normalisation gain
        return 1.0 + 0.1 * torch.randn(shape, generator=gen)
    return 0.1 * torch.randn(shape, generator=gen)  # biases


def synth_state_dict(shapes: dict, seed: int) -> dict:
    r"""shapes: name -> shape (only floating-point parameters).  Names are visited in sorted
    order so the result does not depend on module construction order."""
    gen = torch.Generator().manual_seed(seed)
    return {k: synth_tensor(k, shapes[k], gen) for k in sorted(shapes)}


def shapes_of(state_dict: dict, skip=("sigmas",)) -> dict:
    return {
        k: tuple(v.shape)
        for k, v in state_dict.items()
        if torch.is_floating_point(v) and k.split(".")[-1] not in skip
    }
