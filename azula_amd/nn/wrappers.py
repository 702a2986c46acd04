r"""Time-conditioning wrapper for modulated backbones.

``KarrasDenoiser`` calls ``backbone(c_in * x_t, c_time)`` (reference ``azula/denoise.py:316-320``)
while ``UNet`` / ``ViT`` take a modulation *vector* ``mod`` (``azula/nn/unet.py:205-210``).  The
reference bridges the two with a user-side wrapper (``docs/tutorials/mnist.ipynb``, cell 8):

    mod = Linear(1, D) -> SiLU -> Linear(D, D) applied to c_time[..., None];  net(x_t, mod)

:class:`TimeModulated` is that wrapper with the same parameter names (``time_embedding.{0,2}``
and the wrapped net under ``name``), executed by the small-M HIP linear kernel, and it exposes
the compiled program the fused sampler captures into its per-step graph.
"""

from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from .. import _lib
from ..engine import Tape

__all__ = ["TimeModulated"]


class TimeModulated(nn.Module):
    r"""``forward(x_t, c_time) = net(x_t, time_embedding(c_time[..., None]))``.

    Arguments:
        net: A modulated backbone (``azula_amd.nn.UNet`` / ``ViT``) with ``mod_features == features``.
        features: The number of modulating features D.
        name: Attribute (and ``state_dict`` prefix) under which ``net`` is registered.
    """

    def __init__(self, net: nn.Module, features: int, name: str = "net") -> None:
        super().__init__()
        setattr(self, name, net)
        self._net_name = name
        self.features = features
        self.time_embedding = nn.Sequential(nn.Linear(1, features), nn.SiLU(), nn.Linear(features, features))

    @property
    def net(self) -> nn.Module:
        return getattr(self, self._net_name)

    def _f32(self) -> list[Tensor]:
        r"""fp32 views of the embedding MLP's parameters: the parameters themselves, or (module cast to half
        precision) up-converted copies refreshed when a parameter changes."""
        ps = [self.time_embedding[0].weight, self.time_embedding[0].bias, self.time_embedding[2].weight, self.time_embedding[2].bias]
        if ps[0].dtype == torch.float32:
            return [p.detach() for p in ps]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_f32_key", None) != key:
            self._f32_cache, self._f32_key = [p.detach().float().contiguous() for p in ps], key
        return self._f32_cache

    def _embed(self, tape_or_none, c_time_buf: Tensor, rows: int, mod_buf: Tensor, hid_buf: Tensor):
        w0, b0, w2, b2 = self._f32()
        D = self.features
        args0 = (hid_buf.data_ptr(), D, c_time_buf.data_ptr(), 1, w0.data_ptr(), b0.data_ptr(), rows, D, 1, 0, 1)
        args2 = (mod_buf.data_ptr(), D, hid_buf.data_ptr(), D, w2.data_ptr(), b2.data_ptr(), rows, D, D, 0, 0)
        if tape_or_none is None:
            s = _lib.stream_ptr()
            _lib.call("az_linear_small_f32", *args0, s)
            _lib.call("az_linear_small_f32", *args2, s)
        else:
            tape_or_none.add("az_linear_small_f32", *args0)
            tape_or_none.add("az_linear_small_f32", *args2)

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x_t: Tensor, c_time: Tensor, **kwargs) -> Tensor:
        if not x_t.is_cuda:
            raise RuntimeError("azula_amd backbones execute only on an AMD GPU (no CPU fallback)")
        ct = c_time.to(device=x_t.device, dtype=torch.float32).reshape(-1, 1).contiguous()
        rows = ct.shape[0]
        hid = torch.empty(rows, self.features, dtype=torch.float32, device=x_t.device)
        mod = torch.empty(rows, self.features, dtype=torch.float32, device=x_t.device)
        self._embed(None, ct, rows, mod, hid)
        return self.net(x_t, mod[0] if c_time.ndim == 0 else mod, **kwargs)

    # -- fused sampling ---------------------------------------------------------------------------
    def _az_compile(self, x: Tensor, kwargs: dict, cur_coef: Tensor):
        inner = getattr(self.net, "_az_compile_modulated", None)
        if inner is None or kwargs:
            return None
        dev = x.device
        tape = Tape()
        ct = torch.empty(1, 1, dtype=torch.float32, device=dev)
        hid = torch.empty(1, self.features, dtype=torch.float32, device=dev)
        tape.add("az_coef_c_time_f32", ct.data_ptr(), cur_coef.data_ptr())
        res = inner(x, mod_rows=1)
        if res is None or res[0] is None:
            return None  # unmodulated net, non-image input, cond_channels: the sampler takes the generic loop
        program, mod_buf = res
        self._embed(tape, ct, 1, mod_buf, hid)
        tape.keep.extend([ct, hid, *self._f32()])
        tape.extend(program.tape)
        program.tape = tape
        return program
