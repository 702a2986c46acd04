r"""U-Net backbone executed by hand-written gfx950 kernels.

Drop-in for ``azula.nn.unet`` (reference ``azula/nn/unet.py:18-259``): same constructor
arguments, same ``state_dict`` keys (``descent.{l}.{j}...``, ``ascent.{k}.{j}...``),
same ``forward(x, mod, cond)`` semantics for 2-D inputs.  The modules below only HOLD the
parameters; the forward is a compiled tape of C-ABI kernels on a channel-padded NHWC layout:

* every conv is the MFMA implicit GEMM (``az_conv2d_f32``); SiLU, bias, the gated residual
  ``x + c*y`` (``unet.py:93``), the skip concat ``cat((y, x))`` (``unet.py:257``), nearest x2
  upsampling (``unet.py:187``), ``narrow`` (``unet.py:253-255``) and zero padding are folded into
  its gather / epilogue and never touch HBM as separate passes;
* normalisation + AdaZero modulation ``(a + 1) * norm(x) + b`` (``unet.py:91``) is one stats pass
  + one fused scale/shift pass;
* the first conv reads the NHWC input written by the sampler's transition kernel and the last
  conv writes planar NCHW directly.
"""

from __future__ import annotations

from collections.abc import Sequence

import torch
import torch.nn as nn
from torch import Tensor

from ..engine import Act, Builder, Tape, ada_zero_triple, mod_front_tape, pad4
from .. import _lib, engine

__all__ = ["UNet", "UNetBlock"]


def _conv_holder(cin: int, cout: int, kernel_size, stride=1, identity_init: bool = False) -> nn.Module:
    r"""Parameter holder with the reference's shapes: ``Conv1d`` / ``Conv2d`` / ``Conv3d`` for spatial = 1 / 2 / 3 (``ConvNd``,
    ``azula/nn/layers.py:25-68``); ``len(kernel_size)`` is the number of spatial dimensions."""
    Conv = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[len(kernel_size)]
    conv = Conv(cin, cout, kernel_size=tuple(kernel_size), stride=stride, padding=tuple(k // 2 for k in kernel_size))
    if identity_init:  # azula/nn/layers.py:53-66: near-identity down/up-sampling convolutions
        center = [k // 2 for k in conv.weight.shape[2:]]
        eye = torch.zeros_like(conv.weight.data[:cin])
        for i in range(min(cin, cout)):
            eye[(i, i, *center)] = 1.0
        conv.weight.data[:cin].mul_(1e-2)
        conv.weight.data[:cin].add_(eye)
    return conv


class UNetBlock(nn.Module):
    r"""Parameter holder of a modulated U-Net block (reference ``azula/nn/unet.py:18-116``).

    ``ada_zero`` = Linear(D, D) -> SiLU -> Linear(D, 3C) (last weight x 1e-2), or a raw
    ``(3, C, 1, 1)`` parameter when ``mod_features == 0``;  ``ffn`` = conv -> SiLU -> conv.
    """

    def __init__(
        self,
        channels: int,
        mod_features: int = 0,
        norm: str = "layer",
        groups: int = 16,
        ffn_factor: int = 1,
        spatial: int = 2,
        dropout: float | None = None,
        checkpointing: bool = False,
        kernel_size: Sequence[int] = (3, 3),
        **kwargs,
    ) -> None:
        super().__init__()
        if spatial not in (1, 2, 3):
            raise NotImplementedError("azula_amd.nn.UNet implements spatial = 1, 2 and 3")
        if isinstance(kernel_size, int):  # standalone use passes ConvNd's keyword arguments (reference unet.py:76-83)
            kernel_size = (kernel_size,) * spatial
        kernel_size = tuple(kernel_size)[:spatial] if len(kernel_size) >= spatial else tuple(kernel_size)
        if any(k % 2 == 0 for k in kernel_size) or kwargs.get("stride", 1) not in (1, (1,), [1], (1, 1), [1, 1], (1, 1, 1), [1, 1, 1]):
            raise NotImplementedError("odd kernel sizes with stride 1 only")
        pad = kwargs.get("padding", tuple(k // 2 for k in kernel_size))
        if tuple([pad] * len(kernel_size) if isinstance(pad, int) else pad) != tuple(k // 2 for k in kernel_size):
            raise NotImplementedError("'same' padding (kernel_size // 2) only")
        if norm not in ("layer", "rms", "group"):
            raise NotImplementedError(norm)
        self.periodic = kwargs.get("padding_mode", "zeros") == "circular"
        if kwargs.get("padding_mode", "zeros") not in ("zeros", "circular"):
            raise NotImplementedError(f"padding_mode {kwargs['padding_mode']!r}: zeros and circular only")
        self.channels, self.mod_features, self.spatial = channels, mod_features, spatial
        self.norm_kind, self.groups = norm, min(groups, channels)
        self.ffn_factor = ffn_factor
        if mod_features > 0:
            self.ada_zero = nn.Sequential(
                nn.Linear(mod_features, mod_features), nn.SiLU(), nn.Linear(mod_features, 3 * channels), nn.Identity()
            )
            self.ada_zero[-2].weight.data.mul_(1e-2)
        else:
            self.ada_zero = nn.Parameter(torch.randn(3, channels, *(1,) * spatial))
            self.ada_zero.data.mul_(1e-2)
        self.ffn = nn.Sequential(
            _conv_holder(channels, ffn_factor * channels, kernel_size),
            nn.SiLU(),
            nn.Identity() if dropout is None else nn.Dropout(dropout),
            _conv_holder(ffn_factor * channels, channels, kernel_size),
        )

        self._plans: dict = {}

    def _emit(self, bld: Builder, x: Act, D: int, mod_rows: int, mod_jobs: list, keep_input: bool = False) -> Act:
        r"""y = (a + 1) norm(x) + b;  y = conv(silu(conv(y)));  out = x + c y   (reference ``unet.py:85-95``): one
        statistics + one scale/shift pass, two convolutions with SiLU / gate / residual in their epilogues."""
        Cc, cs = self.channels, pad4(self.channels)
        abc, bstride = ada_zero_triple(bld, self.ada_zero, Cc, D, mod_rows, mod_jobs)
        if self.norm_kind == "group":
            n_ = bld.group_norm(x, self.groups, scale=abc, shift=abc, scale_off=0, shift_off=cs, bstride=bstride, lazy=True)
        else:
            n_ = bld.row_norm(x, 0 if self.norm_kind == "layer" else 1, scale=abc, shift=abc, scale_off=0, shift_off=cs, bstride=bstride)
        c0, c3 = self.ffn[0], self.ffn[3]
        h1 = bld.conv(n_, bld.pack_conv(c0.weight, c0.bias), c0.out_channels, act=1, periodic=self.periodic)
        bld.free(n_)
        y = bld.conv(h1, bld.pack_conv(c3.weight, c3.bias), c3.out_channels, gate=abc, gate_off=2 * cs, gate_bstride=bstride,
                     res=x, periodic=self.periodic, gn_stats=self.norm_kind == "group")  # feeds the next block's norm
        bld.free(h1)
        if not keep_input:
            bld.free(x)
        return y

    def _forward_3d(self, x: Tensor, mod: Tensor | None, out_dtype) -> Tensor:
        r"""(B, C, D, H, W): one-block plan on the volume form (see ``unet3d.py``)."""
        from .unet3d import Vol, block3d

        assert x.ndim == 5 and x.shape[1] == self.channels
        B, Cc, Dd, H, W = x.shape
        D = self.mod_features
        rows = 0
        if D > 0:
            assert mod is not None, "this block is modulated: pass mod"
            rows = 1 if mod.ndim == 1 else mod.shape[0]
            assert rows in (1, B)
        key = (B, Dd, H, W, rows, str(x.device))
        versions = tuple((p.data_ptr(), p._version, p.dtype) for p in self.parameters())
        plan = self._plans.get(key)
        if plan is None or plan[0] != versions:
            bld = Builder(x.device, half=next(self.parameters()).dtype)
            xa = bld.new_act(B * Dd, H, W, Cc, pinned=True)
            xin = Vol(xa.buf, B, Dd, H, W, Cc, xa.cs)
            mod_buf = torch.empty(max(rows, 1), max(D, 1), dtype=torch.float32, device=x.device)
            jobs: list = []
            out = block3d(self, bld, xin, D, rows, jobs, keep_input=True)
            bld.finish()
            tape = bld.tape
            if jobs:
                tape = mod_front_tape(bld, jobs, mod_buf, rows, D)
                tape.extend(bld.tape)
            res = torch.empty(B, Cc, Dd, H, W, dtype=torch.float32, device=x.device)
            tape.add("az_nhwc_to_nchw_f32", res.data_ptr(), out.buf.data_ptr(), B, Cc, Dd * H * W, out.cs)
            plan = (versions, xin, mod_buf, tape, res)
            self._plans.clear()
            self._plans[key] = plan
        _, xin, mod_buf, tape, res = plan
        xs = x.to(torch.float32).contiguous()
        _lib.call("az_nchw_to_nhwc_f32", xin.buf.data_ptr(), xs.data_ptr(), None, B, Cc, Dd * H * W, xin.cs, _lib.stream_ptr())
        if rows:
            mod_buf.copy_(mod.to(torch.float32).reshape(rows, -1))
        tape.run()
        return res.to(out_dtype, copy=True)

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, mod: Tensor | None = None) -> Tensor:
        r"""x: (B, C, H, W) [(B, C, L) for spatial = 1]; mod: (D) or (B, D) -> like x (reference ``unet.py:97-116``)."""
        from .utils import backbone_io_dtype

        out_dtype = backbone_io_dtype(self, x, "azula_amd.nn.UNetBlock")
        if self.spatial == 1:  # (B, C, L): the same kernels and the same plan cache on a one-row image (no module state is touched)
            assert x.ndim == 3
            return self._forward_2d(x[:, :, None], mod, out_dtype)[:, :, 0]
        if self.spatial == 3:
            return self._forward_3d(x, mod, out_dtype)
        return self._forward_2d(x, mod, out_dtype)

    def _forward_2d(self, x: Tensor, mod: Tensor | None, out_dtype) -> Tensor:
        assert x.ndim == 4 and x.shape[1] == self.channels
        B, Cc, H, W = x.shape
        D = self.mod_features
        rows = 0
        if D > 0:
            assert mod is not None, "this block is modulated: pass mod"
            rows = 1 if mod.ndim == 1 else mod.shape[0]
            assert rows in (1, B)
        key = (B, H, W, rows, str(x.device))
        versions = tuple((p.data_ptr(), p._version, p.dtype) for p in self.parameters())
        plan = self._plans.get(key)
        if plan is None or plan[0] != versions:
            bld = Builder(x.device, half=next(self.parameters()).dtype)
            xin = bld.new_act(B, H, W, Cc, pinned=True)
            mod_buf = torch.empty(max(rows, 1), max(D, 1), dtype=torch.float32, device=x.device)
            jobs: list = []
            out = self._emit(bld, xin, D, rows, jobs, keep_input=True)
            bld.finish()
            tape = bld.tape
            if jobs:
                tape = mod_front_tape(bld, jobs, mod_buf, rows, D)
                tape.extend(bld.tape)
            res = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
            tape.add("az_nhwc_to_nchw_f32", res.data_ptr(), out.ptr, B, Cc, H * W, out.cs)
            plan = (versions, xin, mod_buf, tape, res)
            self._plans.clear()
            self._plans[key] = plan
        _, xin, mod_buf, tape, res = plan
        xs = x.to(torch.float32).contiguous()
        _lib.call("az_nchw_to_nhwc_f32", xin.ptr, xs.data_ptr(), None, B, Cc, H * W, xin.cs, _lib.stream_ptr())
        if rows:
            mod_buf.copy_(mod.to(torch.float32).reshape(rows, -1))
        tape.run()
        return res.to(out_dtype, copy=True)


def _copy_tape(t):
    c = Tape()
    c.extend(t)
    return c


class UNetPlan:
    r"""Compiled forward for one (batch, H, W, modulation-rows) signature."""

    def __init__(self, net: "UNet", B: int, H: int, W: int, mod_rows: int, device: torch.device) -> None:
        self.B, self.H, self.W = B, H, W
        # a network cast to half precision keeps its activations in HBM in its own type (engine.HALF_ACT) when every kernel on its
        # tape has the typed form: channel counts in multiples of 8, whole GroupNorm groups in 4-channel chunks, power-of-two strides
        sv0 = (net.stride, net.stride) if isinstance(net.stride, int) else tuple(net.stride)
        blocks = [m for m in net.modules() if isinstance(m, UNetBlock)]
        half_act = (net.spatial == 2 and all(c % 8 == 0 for c in net.hid_channels) and all(v in (1, 2, 4, 8, 16) for v in sv0)
                    and all(b.norm_kind != "group" or (b.channels // b.groups) % 4 == 0 for b in blocks)
                    and all(b.ffn[0].out_channels % 8 == 0 and b.channels <= 4096 for b in blocks))
        bld = self.bld = Builder(device, half=next(net.parameters()).dtype, half_act=half_act)
        cin = net.in_channels + net.cond_channels
        D = net.mod_features
        first = net.descent[0][0]
        # <= 4 input channels through a 3 x 3 first convolution: it reads the latent PLANAR (x_in is then the loop's own
        # (B, C, H, W) layout: channel stride 0 in the fused protocol), see Builder.conv_stem
        self.planar = (engine.STEM_PLANAR and net.spatial == 2 and cin <= 4 and tuple(first.weight.shape[2:]) == (3, 3)
                       and first.out_channels % 4 == 0 and bld.half is None)
        if self.planar:
            self.x_in = Act(torch.empty(B * cin * H * W, dtype=torch.float32, device=device), B, H, W, cin, 0, True)
        else:
            self.x_in = Act(torch.zeros(B * H * W * bld.pad(cin), dtype=torch.float32, device=device), B, H, W, cin, bld.pad(cin), True)
        self.mod = torch.empty(max(mod_rows, 1), max(D, 1), dtype=torch.float32, device=device)
        self.mod_rows = mod_rows
        self.out = torch.empty(B, net.out_channels, H, W, dtype=torch.float32, device=device)
        self.versions = net._param_versions()
        L = len(net.hid_blocks)
        stride = net.stride
        # power-of-two strides (<= 16): the decoder's nearest upsampling is a right shift inside the merge convolution's gather;
        # any other stride: a nearest-upsampling pass of its own in front of the merge convolution (reference unet.py:186,250-254)
        sv = (stride, stride) if isinstance(stride, int) else tuple(stride)
        pow2 = all(v in (1, 2, 4, 8, 16) for v in sv)
        up = 0
        if pow2:
            up = stride.bit_length() - 1 if isinstance(stride, int) else tuple(v.bit_length() - 1 for v in stride)

        mod_jobs: list[tuple] = []  # queued modulation MLPs (ada_zero_triple), emitted together at the tape front
        per = net.periodic
        gn = any(isinstance(m, UNetBlock) and m.norm_kind == "group" for m in net.modules())

        def block(blk: UNetBlock, x: Act, keep_input: bool) -> Act:
            return blk._emit(bld, x, D, mod_rows, mod_jobs, keep_input)

        cur = self.x_in
        skips: list[Act] = []
        for i in range(L):
            first = net.descent[i][0]
            if i > 0:
                skips.append(cur)  # output of level i-1 = memory entry (unet.py:226-230)
            if i == 0 and self.planar:
                nxt = bld.conv_stem(cur.buf, B, cin, H, W, bld.pack_conv(first.weight, first.bias), first.out_channels, periodic=per,
                                    gn_stats=gn)
            else:
                nxt = bld.conv(cur, bld.pack_conv(first.weight, first.bias), first.out_channels, stride=stride if i > 0 else 1,
                               periodic=per, gn_stats=gn)
            cur = nxt
            for j in range(net.hid_blocks[i]):
                cur = block(net.descent[i][1 + j], cur, keep_input=False)
        for k in range(L):
            i = L - 1 - k
            mods = net.ascent[k]
            idx = 0
            if i + 1 < L:
                y = skips[i]
                conv = mods[0]
                if not pow2:
                    wide = bld.upsample_nearest(cur, sv[0], sv[1], y.H, y.W)
                    bld.free(cur)
                    cur = wide
                merged = bld.conv(
                    y, bld.pack_conv(conv.weight, conv.bias, cin0=y.C), conv.out_channels, src1=cur, up1=up,
                    hin=y.H, win=y.W, periodic=per, gn_stats=gn,
                )
                bld.free(cur)
                bld.free(y)
                cur = merged
                idx = 1
            for j in range(net.hid_blocks[i]):
                cur = block(mods[idx + j], cur, keep_input=False)
            idx += net.hid_blocks[i]
            if i == 0:
                conv = mods[idx]
                bld.conv(cur, bld.pack_conv(conv.weight, conv.bias), conv.out_channels, dst_nchw=self.out, periodic=per)
                bld.free(cur)
            # i > 0: nearest upsampling is folded into the next level's merge conv (up1 = log2 stride)
        bld.finish()
        self.tape = bld.tape
        if mod_jobs:  # h_i = silu(W0_i mod + b0_i) for all blocks as ONE GEMV; abc_i = W2_i h_i + b2_i as ONE grouped GEMV
            pre = mod_front_tape(bld, mod_jobs, self.mod, mod_rows, D)
            pre.extend(self.tape)
            self.tape = pre


class UNet(nn.Module):
    r"""Modulated U-Net (reference ``azula/nn/unet.py:119-259``), gfx950-native forward.

    Arguments are those of ``azula.nn.unet.UNet``: ``spatial`` 1 (one-row images), 2 or 3 (volumes: every 3-D convolution
    as depth taps of the 2-D kernels, ``unet3d.py``), zero or circular padding, odd kernels, any integer stride per axis (powers of
    two: the upsampling is folded into the merge convolution; others: ``az_upsample_nearest_f32``; volumes: the same in-plane + a gather of whole planes along the depth axis).
    """

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        cond_channels: int = 0,
        hid_channels: Sequence[int] = (64, 128, 256),
        hid_blocks: Sequence[int] = (3, 3, 3),
        kernel_size: int | Sequence[int] = 3,
        stride: int | Sequence[int] = 2,
        spatial: int = 2,
        periodic: bool = False,
        identity_init: bool = False,
        **kwargs,
    ) -> None:
        super().__init__()
        assert len(hid_blocks) == len(hid_channels)
        if spatial not in (1, 2, 3):
            raise NotImplementedError("azula_amd.nn.UNet implements spatial = 1, 2 and 3")
        self.spatial = spatial
        if isinstance(kernel_size, int):
            kernel_size = [kernel_size] * spatial
        if isinstance(stride, int):
            stride = [stride] * spatial
        assert len(kernel_size) == len(stride) == spatial
        if any(k % 2 == 0 for k in kernel_size):
            raise NotImplementedError("odd kernel sizes only (anisotropic allowed)")
        if any(int(s_) != s_ or s_ < 1 for s_ in stride):
            raise ValueError("integer strides >= 1 only")
        self.in_channels, self.out_channels, self.cond_channels = in_channels, out_channels, cond_channels
        self.hid_channels, self.hid_blocks = tuple(hid_channels), tuple(hid_blocks)
        self.stride = stride[0] if len(set(stride)) == 1 else tuple(stride)  # int (isotropic) or one per axis
        self.mod_features = kwargs.get("mod_features", 0)
        self.periodic = bool(periodic)
        if periodic:  # reference unet.py:175-180: every convolution pads circularly
            kwargs["padding_mode"] = "circular"
        ks = tuple(kernel_size)

        self.descent, self.ascent = nn.ModuleList(), nn.ModuleList()
        for i, num_blocks in enumerate(hid_blocks):
            do, up = nn.ModuleList(), nn.ModuleList()
            for _ in range(num_blocks):
                do.append(UNetBlock(hid_channels[i], kernel_size=ks, spatial=spatial, **kwargs))
                up.append(UNetBlock(hid_channels[i], kernel_size=ks, spatial=spatial, **kwargs))
            if i > 0:
                do.insert(0, _conv_holder(hid_channels[i - 1], hid_channels[i], ks, stride=tuple(stride), identity_init=identity_init))
                up.append(nn.Upsample(scale_factor=tuple(float(s) for s in stride), mode="nearest"))
            else:
                do.insert(0, _conv_holder(in_channels + cond_channels, hid_channels[i], ks))
                up.append(_conv_holder(hid_channels[i], out_channels, ks))
            if i + 1 < len(hid_blocks):
                up.insert(0, _conv_holder(hid_channels[i] + hid_channels[i + 1], hid_channels[i], ks, identity_init=identity_init))
            self.descent.append(do)
            self.ascent.insert(0, up)
        self._plans: dict = {}

    # -- plan management -----------------------------------------------------------------------
    def _param_versions(self) -> tuple:
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def plan3d(self, B: int, D: int, H: int, W: int, mod_rows: int, device: torch.device):
        from .unet3d import UNet3DPlan

        key = (B, D, H, W, mod_rows, str(device))
        p = self._plans.get(key)
        if p is None or p.versions != self._param_versions():
            p = UNet3DPlan(self, B, D, H, W, mod_rows, device)
            self._plans[key] = p
        return p

    def plan(self, B: int, H: int, W: int, mod_rows: int, device: torch.device) -> UNetPlan:
        key = (B, H, W, mod_rows, str(device))
        p = self._plans.get(key)
        if p is None or p.versions != self._param_versions():
            p = UNetPlan(self, B, H, W, mod_rows, device)
            self._plans[key] = p
        return p

    def _check_device(self, x: Tensor) -> torch.dtype:
        from .utils import backbone_io_dtype

        return backbone_io_dtype(self, x, "azula_amd.nn.UNet")

    # -- fused sampling (see azula_amd.sample.BackboneProgram) --------------------------------------
    def _program(self, x: Tensor, mod_rows: int):
        from ..sample import BackboneProgram

        self._check_device(x)
        if self.cond_channels or x.ndim != (5 if self.spatial == 3 else 4):
            return None, None
        if self.spatial == 3:  # the loop's layouts only know (B, C, inner): a volume is an image of D H x W pixels
            B, _, Dd, H, W = x.shape
            p = self.plan3d(B, Dd, H, W, mod_rows, x.device)
            # The loop hands over the pre-scaled input in the latent's OWN (planar) layout, so that it launches the flat
            # transition form (every stream read / written once, 16 B per element: 0.72 - 0.76 of the HBM roofline) instead of
            # the image form with its channel-padded second output (0.59 on volumes); one layout pass in front of
            # the first convolution makes the (B, D, H, W, cs) planes the depth-tap launches read.
            x_planar = torch.empty(B * self.in_channels * Dd * H * W, dtype=torch.float32, device=x.device)
            tape = Tape()
            tape.add("az_nchw_to_nhwc_f32", p.x_in.buf.data_ptr(), x_planar.data_ptr(), None, B, self.in_channels, Dd * H * W, p.x_in.cs,
                     keep=[x_planar])
            tape.extend(p.tape)
            prog = BackboneProgram(tape=tape, x_in=x_planar, x_in_cs=0, out=p.out, f_channels=self.out_channels, f_nhwc=False)
            prog.tape.keep.append(p)
            return prog, p.mod
        B, _, H, W = x.shape
        p = self.plan(B, H, W, mod_rows, x.device)
        prog = BackboneProgram(
            tape=_copy_tape(p.tape), x_in=p.x_in.buf, x_in_cs=p.x_in.cs, out=p.out, f_channels=self.out_channels,
            f_nhwc=False,
        )
        prog.tape.keep.append(p)
        return prog, p.mod

    def _az_compile_modulated(self, x: Tensor, mod_rows: int = 1):
        r"""(program, mod buffer): the caller's tape must fill the (mod_rows, D) buffer first."""
        if self.mod_features == 0:
            return None
        return self._program(x, mod_rows)

    def _az_compile(self, x: Tensor, kwargs: dict, cur_coef: Tensor):
        r"""Bare UNet as a KarrasDenoiser backbone: only meaningful without modulation (the
        second positional argument, c_time, is then ignored exactly as in the reference)."""
        if self.mod_features > 0 or kwargs:
            return None
        return self._program(x, 0)[0]

    def _forward_3d(self, x: Tensor, mod: Tensor | None, out_dtype) -> Tensor:
        r"""(B, C, D, H, W) volumes: every 3-D convolution as depth taps of the 2-D kernels (``unet3d.py``)."""
        assert x.ndim == 5, "spatial = 3: expected (B, C, D, H, W)"
        B, Cin, D, H, W = x.shape
        assert Cin == self.in_channels + self.cond_channels
        rows = 0
        if self.mod_features > 0:
            assert mod is not None, "this UNet is modulated: pass mod"
            mod = mod.to(torch.float32)
            rows = 1 if mod.ndim == 1 else mod.shape[0]
            assert rows in (1, B)
        p = self.plan3d(B, D, H, W, rows, x.device)
        s = _lib.stream_ptr()
        _lib.call("az_nchw_to_nhwc_f32", p.x_in.buf.data_ptr(), x.data_ptr(), None, B, Cin, D * H * W, p.x_in.cs, s)
        if rows:
            p.mod.copy_(mod.reshape(rows, -1))
        p.tape.run(s)
        return p.out.to(out_dtype, copy=True)

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, mod: Tensor | None = None, cond: Tensor | None = None) -> Tensor:
        r"""x: (B, C_i, H, W); mod: (D) or (B, D); cond: (B, C_c, H, W) -> (B, C_o, H, W).  With ``spatial = 1`` the
        tensors are (B, C, L): a one-row image through the same kernels (the 1-D filters sit in the middle row of
        3 x 3 ones, the other rows only ever meet the padding)."""
        out_dtype = self._check_device(x)
        if cond is not None:
            x = torch.cat((x, cond), dim=1)
        if self.spatial == 1:
            assert x.ndim == 3, "spatial = 1: expected (B, C, L)"
            x = x[:, :, None]
        x = x.to(torch.float32).contiguous()
        if self.spatial == 3:
            return self._forward_3d(x, mod, out_dtype)
        B, Cin, H, W = x.shape
        assert Cin == self.in_channels + self.cond_channels
        if self.mod_features > 0:
            assert mod is not None, "this UNet is modulated: pass mod"
            mod = mod.to(torch.float32)
            rows = 1 if mod.ndim == 1 else mod.shape[0]
            assert rows in (1, B)
        else:
            rows = 0
        p = self.plan(B, H, W, rows, x.device)
        s = _lib.stream_ptr()
        if p.planar:
            p.x_in.buf.copy_(x.reshape(-1))
        else:
            _lib.call("az_nchw_to_nhwc_f32", p.x_in.ptr, x.data_ptr(), None, B, Cin, H * W, p.x_in.cs, s)
        if rows:
            p.mod.copy_(mod.reshape(rows, -1))
        p.tape.run(s)
        out = p.out.to(out_dtype, copy=True)
        return out[:, :, 0] if self.spatial == 1 else out
