r"""DiT / ViT backbones executed by gfx950 kernels -- drop-ins for ``azula.nn.dit`` / ``azula.nn.vit``.

Same constructors and ``state_dict`` keys as the reference (``azula/nn/dit.py:24-218``,
``azula/nn/vit.py:22-108``, ``azula/nn/attention.py:17-108``).  Note the reference's
block is NOT the standard two-branch DiT: ONE (a, b, c) AdaLN-zero triple per block,

    y = (a + 1) * RMSNorm(x) + b;   y = y + MSA(y);   y = FFN(y);   out = x + c * y   (dit.py:102-110)

Compiled forward (token tensors are "NHWC" images of width 1, so every linear is the MFMA
implicit GEMM with ksize = 1):

* patchify is an index remap fused with the ``c_in`` pre-scale; ``in_proj`` adds the
  (batch-shared, precomputed) positional embedding in its epilogue;
* RMSNorm + modulation is one row-norm pass; q/k RMSNorm and the 1/sqrt(C) scale are folded
  into the attention kernel's operand loads; ``y + y_proj(att)``, SiLU and ``x + c * ffn`` are
  GEMM epilogues.
"""

from __future__ import annotations

import math
from collections.abc import Sequence

import torch
import torch.nn as nn
from torch import Tensor

from .. import _lib, engine
from ..engine import Act, Builder, Tape, ada_zero_triple, mod_front_tape, pad4

__all__ = ["DiT", "DiTBlock", "MultiheadSelfAttention", "ViT"]


class MultiheadSelfAttention(nn.Module):
    r"""Parameter holder (reference ``azula/nn/attention.py:17-70``): fused ``qkv_proj`` and
    bias-free ``y_proj``; per-head q/k RMSNorm has no parameters."""

    def __init__(
        self,
        channels: int,
        pos_channels: int = 1,
        attention_heads: int = 1,
        qkv_bias: bool = True,
        qk_norm: bool = True,
        rope: bool = False,
        dropout: float | None = None,
    ) -> None:
        super().__init__()
        assert channels % attention_heads == 0
        self.qkv_proj = nn.Linear(channels, 3 * channels, bias=qkv_bias)
        self.y_proj = nn.Linear(channels, channels, bias=False)
        if rope:  # learned rotary axes (reference attention.py:60-68): random directions, log-uniform magnitudes
            magnitude = torch.exp(math.log(1e-1) * torch.rand(channels // 2, 1))
            direction = torch.randn(channels // 2, pos_channels)
            direction = direction / torch.linalg.norm(direction, dim=-1, keepdim=True)
            self.theta_proj = nn.Linear(pos_channels, channels // 2, bias=False)
            self.theta_proj.weight.data.copy_(magnitude * direction)
        else:
            self.theta_proj = None
        self.heads = attention_heads
        self.qk_norm = qk_norm
        self._plans: dict = {}

    def _emit(self, bld: Builder, y: Act, pos_h: Tensor | None, mask: Tensor | None, res: Act | None = None) -> Act:
        r"""qkv projection -> attention (q/k RMS norm, RoPE, mask in the kernel) -> y_proj (+ ``res``)."""
        C_ = self.qkv_proj.in_features
        rope = None
        if self.theta_proj is not None:  # theta = theta_proj(pos) on the host (reference op order), tables on device
            if pos_h is None:
                raise ValueError("this attention layer uses RoPE: pass pos")
            theta = torch.nn.functional.linear(pos_h.to(torch.float32).cpu(), self.theta_proj.weight.detach().float().cpu())
            rope = (bld.const(torch.cos(theta)), bld.const(torch.sin(theta)))
        # with RoPE: q / k RMS norm and the rotation in the projection's epilogue, once per layer (C6-like: attention 0.38 -> 0.49 of
        # the MFMA peak).  The RMS norm alone stays in the attention kernel: measured on DiT-B/2 (C3) the kernel gains 7 % but the
        # captured step loses 0.4 % (23.77 vs 23.67 ms, two rounds each) -- a norm of the staged K rows is nearly free there.
        d = C_ // self.heads
        dp = engine.attn_padded_dim(d, bld.half)
        if dp != d:  # a head size the kernels are not instantiated for: zero-padded heads (engine.ATTN_HEAD_DIMS; G24)
            if rope is not None:
                theta_p = engine.pad_head_table(theta, self.heads, d // 2, dp // 2)
                rope = (bld.const(torch.cos(theta_p)), bld.const(torch.sin(theta_p)))
            wq, bq = engine.pad_qkv_heads(self.qkv_proj.weight, self.qkv_proj.bias, self.heads, d, dp, "nHC")
            qkv = bld.conv(y, bld.pack_conv(wq, bq), 3 * self.heads * dp)
            att = bld.attention(qkv, self.heads, "nHC", self.qk_norm, 1.0 / math.sqrt(d), rope=rope, mask=mask, norm_dim=d)
            bld.free(qkv)
            out = bld.conv(att, bld.pack_conv(engine.pad_proj_heads(self.y_proj.weight, self.heads, d, dp), None), C_, res=res)
            bld.free(att)
            return out
        prep = dict(heads=self.heads, head_dim=d, rmsnorm=self.qk_norm, eps=1e-5, rope=rope) if rope else None
        qkv = bld.conv(y, bld.pack_conv(self.qkv_proj.weight, self.qkv_proj.bias), 3 * C_, qk_prep=prep)
        att = bld.attention(qkv, self.heads, "nHC", self.qk_norm, 1.0 / math.sqrt(d), rope=rope, mask=mask)
        bld.free(qkv)
        out = bld.conv(att, bld.pack_conv(self.y_proj.weight, None), C_, res=res)
        bld.free(att)
        return out

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, pos: Tensor | None = None, mask: Tensor | None = None) -> Tensor:
        r"""x: (*, L, H C) tokens; pos: (L, P) positions (RoPE only); mask: boolean (L, L) or broadcastable to
        (B, H, L, L), True = attend -> (*, L, H C)   (reference ``azula/nn/attention.py:72-110``)."""
        from .utils import backbone_io_dtype

        out_dtype = backbone_io_dtype(self, x, "azula_amd.nn.MultiheadSelfAttention")
        plan = _token_plan(self, x, pos, mask, 0, lambda bld, xin, pos_h, plan: self._emit(bld, xin, pos_h, mask))
        return plan.run(x, None).to(out_dtype)


def _half_act_ok(module: nn.Module) -> bool:
    r"""A token module cast to half precision can keep its activations in HBM in its own type (engine.HALF_ACT) when every
    kernel on its tape has the typed form: token widths that are multiples of 8 and at most 4096 (az_rownorm_mod_h16, 16-byte
    vectors of 8 values), SwiGLU only through the GEMM's epilogue."""
    for m in module.modules():
        if isinstance(m, MultiheadSelfAttention):
            c = m.qkv_proj.in_features
            if c % 8 or c > 4096 or (c // m.heads) % 4:
                return False
        if isinstance(m, DiTBlock):
            if m.channels % 8 or m.channels > 4096 or m.ffn[3].in_features % 8:
                return False
            if m.ffn_activation == "swiglu" and m.ffn[0].out_features % 16:
                return False
    return True


class _TokenPlan:
    r"""One-module plan on a (B, L, C) token tensor: input Act, optional modulation front, output Act."""

    def __init__(self, module: nn.Module, B: int, L: int, Cin: int, pos_h, mod_rows: int, D: int, emit, device) -> None:
        bld = self.bld = Builder(device, half=next(module.parameters()).dtype, half_act=_half_act_ok(module) and Cin % 8 == 0)
        self.versions = _versions(module)
        self.x_in = bld.new_act(B, L, 1, Cin, pinned=True)
        self.mod = torch.empty(max(mod_rows, 1), max(D, 1), dtype=torch.float32, device=device)
        self.mod_jobs: list = []
        self.mod_rows = mod_rows
        self.out = emit(bld, self.x_in, pos_h, self)
        bld.finish()
        self.tape = bld.tape
        if self.mod_jobs:
            pre = mod_front_tape(bld, self.mod_jobs, self.mod, mod_rows, D)
            pre.extend(self.tape)
            self.tape = pre

    def run(self, x: Tensor, mod: Tensor | None) -> Tensor:
        t, (B, L) = self.x_in, (self.x_in.B, self.x_in.H)
        xf = x.to(torch.float32).reshape(B * L, -1)
        if t.cs == xf.shape[1]:
            t.buf.copy_(xf.reshape(-1))
        else:
            t.buf.zero_()
            t.buf.view(B * L, t.cs)[:, : xf.shape[1]].copy_(xf)
        if self.mod_rows:
            self.mod.copy_(mod.to(torch.float32).reshape(self.mod_rows, -1))
        self.tape.run()
        o = self.out
        return o.buf.view(B, L, o.cs)[..., : o.C].reshape(*x.shape[:-1], o.C).clone()


def _versions(module: nn.Module) -> tuple:
    return tuple((p.data_ptr(), p._version, p.dtype) for p in module.parameters())


def _token_plan(module, x: Tensor, pos, mask, D: int, emit, mod: Tensor | None = None) -> _TokenPlan:
    r"""Plan cache of a standalone token module, keyed on shapes, positions, mask content and parameter versions."""
    assert x.ndim >= 2, "expected (*, L, C) tokens"
    L, Cin = x.shape[-2], x.shape[-1]
    B = x.numel() // (L * Cin)
    rows = 0
    if D > 0:
        assert mod is not None, "this block is modulated: pass mod"
        rows = 1 if mod.ndim == 1 else mod.numel() // mod.shape[-1]
        assert rows in (1, B), "mod must be (D) or (*, D) with the leading shape of x"
    # Positions and mask are BAKED into the plan (RoPE tables, mask bytes).  The key holds their CONTENT (bytes compare by
    # value: no hash-collision reuse); a tensor already seen -- same storage, shape and version counter -- is recognised
    # without copying it to the host again, so a steady-state forward does not synchronise the device.
    def ident(t):
        return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), str(t.dtype), str(t.device))

    seen = module.__dict__.setdefault("_plan_seen", {})
    ik = (ident(pos), ident(mask))
    hit = seen.get(ik)  # (content, pos, mask): the tensors are kept alive, so their storage cannot be handed to others
    content = hit[0] if hit is not None and hit[1] is pos and hit[2] is mask else None
    pos_h = None
    if content is None:
        pos_h = None if pos is None else pos.detach().to("cpu", torch.float32).reshape(L, -1)
        content = (None if pos_h is None else pos_h.numpy().tobytes(),
                   None if mask is None else (tuple(mask.shape), mask.detach().cpu().numpy().tobytes()))
        if len(seen) >= 2:
            seen.clear()
        seen[ik] = (content, pos, mask)
    key = (B, L, Cin, rows, str(x.device), content)
    plan = module._plans.get(key)
    if plan is None or plan.versions != _versions(module):
        if plan is not None or len(module._plans) >= 4:
            module._plans.clear()  # stale parameters, or more than a few live (positions, mask) variants
        if pos_h is None and pos is not None:
            pos_h = pos.detach().to("cpu", torch.float32).reshape(L, -1)
        plan = _TokenPlan(module, B, L, Cin, pos_h, rows, D, emit, x.device)
        module._plans[key] = plan
    return plan


class DiTBlock(nn.Module):
    r"""Parameter holder of a modulated DiT block (reference ``azula/nn/dit.py:24-93``)."""

    def __init__(
        self,
        channels: int,
        mod_features: int = 0,
        ffn_factor: int = 4,
        ffn_activation: str = "silu",
        dropout: float | None = None,
        checkpointing: bool = False,
        **kwargs,
    ) -> None:
        super().__init__()
        if ffn_activation not in ("relu", "relu2", "silu", "swiglu"):
            raise NotImplementedError(f"Unknown activation '{ffn_activation}'.")
        self.channels, self.mod_features, self.ffn_activation = channels, mod_features, ffn_activation
        if mod_features > 0:
            self.ada_zero = nn.Sequential(
                nn.Linear(mod_features, mod_features), nn.SiLU(), nn.Linear(mod_features, 3 * channels), nn.Identity()
            )
            self.ada_zero[-2].weight.data.mul_(1e-2)
        else:
            self.ada_zero = nn.Parameter(torch.randn(3, channels))
            self.ada_zero.data.mul_(1e-2)
        self.msa = MultiheadSelfAttention(channels, **kwargs)
        self.ffn = nn.Sequential(
            nn.Linear(channels, ffn_factor * channels),
            nn.Identity(),  # activation (parameter-free; applied in the GEMM epilogue / az_swiglu_f32)
            nn.Identity() if dropout is None else nn.Dropout(dropout),
            nn.Linear(ffn_factor * channels // (2 if ffn_activation == "swiglu" else 1), channels),
        )
        self._plans: dict = {}

    def _emit(self, bld: Builder, x: Act, pos_h, mask, D: int, mod_rows: int, mod_jobs: list, keep_input: bool = False) -> Act:
        r"""y = (a + 1) RMSNorm(x) + b;  y = y + MSA(y);  y = FFN(y);  out = x + c y   (reference ``dit.py:95-112``)."""
        C_, cs = self.channels, pad4(self.channels)
        abc, bstride = ada_zero_triple(bld, self.ada_zero, C_, D, mod_rows, mod_jobs)
        y = bld.row_norm(x, 1, scale=abc, shift=abc, scale_off=0, shift_off=cs, bstride=bstride)
        y2 = self.msa._emit(bld, y, pos_h, mask, res=y)
        bld.free(y)
        f0, f3 = self.ffn[0], self.ffn[3]
        code = {"silu": 1, "relu": 2, "relu2": 3, "swiglu": 0}[self.ffn_activation]
        fused_glu = self.ffn_activation == "swiglu" and f0.out_features % 8 == 0  # SwiGLU in the GEMM's epilogue (act = 4)
        f1 = bld.conv(y2, bld.pack_conv(f0.weight, f0.bias), f0.out_features, act=4 if fused_glu else code)
        bld.free(y2)
        if self.ffn_activation == "swiglu" and not fused_glu:  # x1 * silu(x2) over interleaved pairs (layers.py:107-110)
            glu = bld.new_act(f1.B, f1.H, f1.W, f1.C // 2)
            bld.tape.add("az_swiglu_f32", glu.ptr, f1.ptr, f1.B * f1.H * f1.W, f1.C // 2, f1.cs, glu.cs)
            bld.free(f1)
            f1 = glu
        out = bld.conv(f1, bld.pack_conv(f3.weight, f3.bias), C_, gate=abc, gate_off=2 * cs, gate_bstride=bstride, res=x)
        bld.free(f1)
        if not keep_input:
            bld.free(x)
        return out

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, mod: Tensor | None = None, pos: Tensor | None = None, mask: Tensor | None = None) -> Tensor:
        r"""x: (*, L, C); mod: (D) or (*, D); pos: (L, P); mask: boolean, broadcastable to (B, H, L, L) -> (*, L, C)
        (reference ``azula/nn/dit.py:114-133``)."""
        from .utils import backbone_io_dtype

        out_dtype = backbone_io_dtype(self, x, "azula_amd.nn.DiTBlock")
        D = self.mod_features

        def emit(bld, xin, pos_h, plan):
            return self._emit(bld, xin, pos_h, mask, D, plan.mod_rows, plan.mod_jobs, keep_input=True)

        plan = _token_plan(self, x, pos, mask, D, emit, mod)
        return plan.run(x, mod).to(out_dtype)


def host_sine_encoding(x: Tensor, features: int, omega: float) -> Tensor:
    r"""sin block || cos block, computed on the host in the reference's op order
    (``azula/nn/layers.py:286-299``)."""
    x = x.unsqueeze(dim=-1)
    freqs = torch.linspace(0, 1, features // 2, dtype=x.dtype)
    freqs = torch.exp(math.log(1 / omega) * freqs)
    return torch.cat((torch.sin(x * freqs), torch.cos(x * freqs)), dim=-1)


class DiTPlan:
    r"""Compiled forward for (batch, tokens[, patch geometry], modulation rows)."""

    def __init__(self, net: "DiT", B: int, L: int, pos: Tensor, mod_rows: int, device, patch=None) -> None:
        bld = self.bld = Builder(device, half=next(net.parameters()).dtype, half_act=_half_act_ok(net) and net.hid_channels % 8 == 0)
        D, C_ = net.mod_features, net.hid_channels
        self.mod = torch.empty(max(mod_rows, 1), max(D, 1), dtype=torch.float32, device=device)
        self.mod_rows = mod_rows
        self.versions = net._param_versions()
        cin = net.in_proj.in_features
        cout = net.out_proj.out_features
        if patch is not None:
            Z, H, W, p, pu = patch
            self.x_nchw = torch.empty(B, Z, H, W, dtype=torch.float32, device=device)
            tokens = bld.new_act(B, L, 1, cin, pinned=True, f32=True)  # (the plan's input stays fp32: the first GEMM rounds it per tile)
            bld.tape.add("az_patchify_f32", tokens.ptr, self.x_nchw.data_ptr(), None, B, Z, H, W, p, tokens.cs)
            self.x_in_buf, self.x_in_cs = self.x_nchw, 0
        else:
            tokens = bld.new_act(B, L, 1, cin, pinned=True, f32=True)
            self.x_in_buf, self.x_in_cs = tokens.buf, tokens.cs
        self.tokens = tokens

        # positional embedding table (L, C): host sine encoding -> one GEMM at build time
        enc = host_sine_encoding(pos.to(torch.float32).cpu(), C_, omega=1e2).flatten(-2)  # (L, P*C)
        pb = Builder(device)
        enc_act = Act(enc.to(device).contiguous().reshape(-1), 1, L, 1, enc.shape[1], enc.shape[1], True)
        ptab = pb.conv(enc_act, pb.pack_conv(net.pos_embedding[2].weight, None), C_)
        pb.finish()
        pb.tape.run()
        if bld.half_act:  # the residual operand of the first GEMM has the destination's element type
            pt = torch.zeros(L, bld.pad(C_), dtype=bld.half, device=device)
            pt[:, :C_] = ptab.buf.view(L, ptab.cs)[:, :C_]
            ptab = Act(pt.reshape(-1), 1, L, 1, C_, bld.pad(C_), True)
        self.pos_table = ptab
        bld.tape.keep.extend([pb, enc_act])
        ptab.pinned = True

        x = bld.conv(tokens, bld.pack_conv(net.in_proj.weight, net.in_proj.bias), C_, res=_bcast(ptab))
        mod_jobs: list[tuple] = []  # queued modulation MLPs of the blocks, emitted together at the tape front
        for blk in net.blocks:
            x = blk._emit(bld, x, pos, None, D, mod_rows, mod_jobs)
        o = bld.conv(x, bld.pack_conv(net.out_proj.weight, net.out_proj.bias), cout, out_f32=True)  # (the plan's output: fp32)
        bld.free(x)
        if patch is not None:  # '... A B (Z a b) -> ... Z (A a) (B b)' with the UNPATCH size (reference vit.py:63-74,104-106)
            Z, H, W, p, pu = patch
            Zo, Ho, Wo = cout // (pu * pu), H // p * pu, W // p * pu
            self.out = torch.empty(B, Zo, Ho, Wo, dtype=torch.float32, device=device)
            bld.tape.add("az_unpatchify_f32", self.out.data_ptr(), o.ptr, B, Zo, Ho, Wo, pu, o.cs)
            self.out_tokens = None
        else:
            self.out, self.out_tokens = None, o
        bld.finish()
        self.tape = bld.tape
        if mod_jobs:  # all blocks' modulation MLPs read only `mod`: one GEMV + one grouped GEMV at the front
            pre = mod_front_tape(bld, mod_jobs, self.mod, mod_rows, D)
            pre.extend(self.tape)
            self.tape = pre


class _Bcast:
    r"""Marks a residual Act as batch-shared (conv epilogue ``res_bcast``)."""

    def __init__(self, act: Act) -> None:
        self.act = act


def _bcast(a: Act) -> "_Bcast":
    return _Bcast(a)


class DiT(nn.Module):
    r"""Modulated DiT-like module on token tensors (reference ``azula/nn/dit.py:135-218``)."""

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        cond_channels: int = 0,
        mod_features: int = 0,
        pos_channels: int = 1,
        hid_channels: int = 1024,
        hid_blocks: int = 3,
        **kwargs,
    ) -> None:
        super().__init__()
        self.mod_features, self.hid_channels, self.pos_channels = mod_features, hid_channels, pos_channels
        self.cond_channels = cond_channels
        self.in_proj = nn.Linear(in_channels + cond_channels, hid_channels)
        self.out_proj = nn.Linear(hid_channels, out_channels)
        self.pos_embedding = nn.Sequential(
            nn.Identity(), nn.Identity(), nn.Linear(pos_channels * hid_channels, hid_channels, bias=False)
        )
        self.pos_embedding[-1].weight.data.mul_(1e-2)
        self.blocks = nn.ModuleList([
            DiTBlock(channels=hid_channels, pos_channels=pos_channels, mod_features=mod_features, **kwargs)
            for _ in range(hid_blocks)
        ])
        self._plans: dict = {}

    def _param_versions(self) -> tuple:
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def _check_device(self, x: Tensor) -> torch.dtype:
        from .utils import backbone_io_dtype

        return backbone_io_dtype(self, x, f"azula_amd.nn.{type(self).__name__}")

    def _mod_rows(self, mod, B: int) -> int:
        if self.mod_features == 0:
            return 0
        assert mod is not None, "this network is modulated: pass mod"
        rows = 1 if mod.ndim == 1 else mod.shape[0]
        assert rows in (1, B)
        return rows

    def _get_plan(self, key, make):
        p = self._plans.get(key)
        if p is None or p.versions != self._param_versions():
            p = make()
            self._plans[key] = p
        return p

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, mod: Tensor | None = None, pos: Tensor | None = None, cond: Tensor | None = None):
        r"""x: (B, L, C_i) tokens -> (B, L, C_o).  ``pos``: (L, P) or None (sequence indices)."""
        out_dtype = self._check_device(x)
        if cond is not None:
            x = torch.cat((x, cond), dim=-1)
        assert x.ndim == 3, "DiT.forward expects (B, L, C) tokens"
        x = x.to(torch.float32).contiguous()
        B, L, Cin = x.shape
        rows = self._mod_rows(mod, B)
        if pos is None:
            pos_h = torch.arange(L, dtype=torch.float32)[:, None]
            pkey = None
        else:
            pos_h = pos.detach().to("cpu", torch.float32).reshape(L, -1)
            pkey = hash(pos_h.numpy().tobytes())
        plan = self._get_plan((B, L, rows, pkey, str(x.device)), lambda: DiTPlan(self, B, L, pos_h, rows, x.device))
        t = plan.tokens
        if t.cs == Cin:
            t.buf.copy_(x.reshape(-1))
        else:
            t.buf.zero_()
            t.buf.view(B * L, t.cs)[:, :Cin].copy_(x.reshape(B * L, Cin))
        if rows:
            plan.mod.copy_(mod.to(torch.float32).reshape(rows, -1))
        plan.tape.run()
        o = plan.out_tokens
        return o.buf.view(B, L, o.cs)[..., : o.C].to(out_dtype, copy=True)


def _patchify_nd(x: Tensor, patch: Sequence[int]) -> Tensor:
    r"""'B Z (A a) (B b) ... -> B A B ... (Z a b ...)'."""
    B, Z, *dims = x.shape
    n = len(patch)
    assert len(dims) == n and all(d % p == 0 for d, p in zip(dims, patch)), (x.shape, patch)
    x = x.reshape(B, Z, *[v for d, p in zip(dims, patch) for v in (d // p, p)])
    grid_axes = [2 + 2 * i for i in range(n)]
    patch_axes = [3 + 2 * i for i in range(n)]
    x = x.permute(0, *grid_axes, 1, *patch_axes)
    return x.reshape(B, *[d // p for d, p in zip(dims, patch)], Z * math.prod(patch))


def _unpatchify_nd(y: Tensor, patch: Sequence[int]) -> Tensor:
    r"""'B A B ... (Z a b ...) -> B Z (A a) (B b) ...'."""
    B, *grid, F = y.shape
    n = len(patch)
    Z = F // math.prod(patch)
    y = y.reshape(B, *grid, Z, *patch)
    order = [0, 1 + n] + [v for i in range(n) for v in (1 + i, 2 + n + i)]
    return y.permute(*order).reshape(B, Z, *[g * p for g, p in zip(grid, patch)])


class ViT(DiT):
    r"""Modulated ViT-like module on images (reference ``azula/nn/vit.py:22-108``)."""

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        cond_channels: int = 0,
        mod_features: int = 0,
        hid_channels: int = 1024,
        hid_blocks: int = 3,
        spatial: int = 2,
        patch_size: int | Sequence[int] = 1,
        unpatch_size: int | Sequence[int] | None = None,
        **kwargs,
    ) -> None:
        if isinstance(patch_size, int):
            patch_size = [patch_size] * spatial
        if unpatch_size is None:
            unpatch_size = patch_size
        elif isinstance(unpatch_size, int):
            unpatch_size = [unpatch_size] * spatial
        assert len(patch_size) == len(unpatch_size) == spatial
        super().__init__(
            in_channels=math.prod(patch_size) * in_channels, out_channels=math.prod(unpatch_size) * out_channels,
            cond_channels=math.prod(patch_size) * cond_channels, mod_features=mod_features, pos_channels=spatial,
            hid_channels=hid_channels, hid_blocks=hid_blocks, **kwargs,
        )
        self.patch_shape, self.unpatch_shape = tuple(patch_size), tuple(unpatch_size)
        # images with square patches take the fused patchify / unpatchify kernels (and the captured sampling graph);
        # any other geometry (1-D, 3-D, anisotropic patches) rearranges with torch index ops around the token network
        self.native = spatial == 2 and len(set(patch_size)) == 1 and len(set(unpatch_size)) == 1
        self.patch_size, self.unpatch_size = patch_size[0], unpatch_size[0]
        self.image_in, self.image_out, self.image_cond = in_channels, out_channels, cond_channels
        self.spatial = spatial

    def _vit_plan(self, B: int, H: int, W: int, rows: int, device) -> DiTPlan:
        p = self.patch_size
        Hp, Wp = H // p, W // p
        pos = torch.cartesian_prod(torch.arange(Hp, dtype=torch.float32), torch.arange(Wp, dtype=torch.float32))
        pos = pos.reshape(-1, 2)
        return self._get_plan(
            ("vit", B, H, W, rows, str(device)),
            lambda: DiTPlan(self, B, Hp * Wp, pos, rows, device, patch=(self.image_in + self.image_cond, H, W, p, self.unpatch_size)),
        )

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, mod: Tensor | None = None, cond: Tensor | None = None) -> Tensor:
        r"""x: (B, C_i, H, W); mod: (D) or (B, D) -> (B, C_o, H, W)."""
        out_dtype = self._check_device(x)
        assert (cond is not None) == (self.image_cond > 0), "pass cond iff the network was built with cond_channels"
        if not self.native:
            return self._forward_rearranged(x, mod, cond)
        B, Z, H, W = x.shape
        assert Z == self.image_in and H % self.patch_size == 0 and W % self.patch_size == 0
        rows = self._mod_rows(mod, B)
        plan = self._vit_plan(B, H, W, rows, x.device)
        # patchify(x) || patchify(cond) along the token features == patchify of the channel concatenation: the
        # feature index is (channel, a, b) with the channel slowest (reference vit.py:92-96, layers.py:198-222)
        plan.x_nchw[:, :Z].copy_(x)
        if cond is not None:
            plan.x_nchw[:, Z:].copy_(cond)
        if rows:
            plan.mod.copy_(mod.to(torch.float32).reshape(rows, -1))
        plan.tape.run()
        return plan.out.to(out_dtype, copy=True)

    def _forward_rearranged(self, x: Tensor, mod, cond) -> Tensor:
        r"""Any number of spatial dimensions / anisotropic patches: ``Patchify`` and ``Unpatchify`` (reference
        ``azula/nn/layers.py:198-247``, channel_last) as torch reshapes around the compiled token network."""
        assert x.ndim == 2 + self.spatial and x.shape[1] == self.image_in
        t = _patchify_nd(x, self.patch_shape)
        grid = t.shape[1:-1]
        pos = torch.cartesian_prod(*(torch.arange(n, dtype=torch.float32) for n in grid)).reshape(-1, len(grid))
        t = t.flatten(1, -2)
        c = None if cond is None else _patchify_nd(cond, self.patch_shape).flatten(1, -2)
        y = DiT.forward(self, t, mod, pos=pos, cond=c)
        return _unpatchify_nd(y.unflatten(1, grid), self.unpatch_shape)

    # -- fused sampling ---------------------------------------------------------------------------
    def _az_compile_modulated(self, x: Tensor, mod_rows: int = 1):
        from ..sample import BackboneProgram
        from .unet import _copy_tape

        if not self.native:
            return None
        if self.mod_features == 0 or x.ndim != 4 or self.image_cond or self.unpatch_size != self.patch_size:
            return None  # cond / a different output geometry: the generic step loop (eager forward per step)
        self._check_device(x)
        B, _, H, W = x.shape
        plan = self._vit_plan(B, H, W, mod_rows, x.device)
        prog = BackboneProgram(
            tape=_copy_tape(plan.tape), x_in=plan.x_nchw, x_in_cs=0, out=plan.out, f_channels=self.image_out, f_nhwc=False
        )
        prog.tape.keep.append(plan)
        return prog, plan.mod
