r"""U-Net on volumes (``spatial = 3``) through the 2-D kernels.

Reference: ``azula/nn/unet.py:119-259`` with ``ConvNd(spatial=3)`` (``azula/nn/layers.py:25-68``), i.e. ``Conv3d`` with
'same' padding, zero or circular.  A volume is stored (B, D, H, W, C) = B * D channel-padded NHWC planes, and a 3-D
convolution is the sum over its depth taps of 2-D convolutions of depth-shifted planes,

    out[b, d] = sum_j conv2d(x[b, s d + j - p], w[:, :, j]),

each tap one launch of the 2-D kernel (Winograd / direct / bf16x3 / half-precision, in-plane padding, stride, nearest
upsampling, concatenation and narrow exactly as for images) over ALL planes of all samples (``AzConvArgs.depth``: a plane whose
tap leaves its volume reads zeros, or -- ``depth_wrap``, circular padding -- the plane at the other end of the volume; round 4:
also for odd depths and half-precision modules) that ACCUMULATES in place through the epilogue's residual operand.  The centre tap runs
first (it exists for every output plane) and carries the bias; a gate distributes over the taps
(``x + c (sum_j v_j + bias) = x + c (v_0 + bias) + c v_1 + ...``), only the SiLU after a block's first convolution needs a
pass of its own.  Norms see a volume as one (D H) x W image.  Depth padding is a skipped (zeros) or wrapped (circular)
plane index -- no padded copies.
"""

from __future__ import annotations

import torch

from ..engine import Act, Builder, ada_zero_triple, mod_front_tape, pad4


class Vol:
    r"""(B, D, H, W, cs) floats in ``buf``; ``C`` real channels."""

    __slots__ = ("buf", "B", "D", "H", "W", "C", "cs", "bounded")

    def __init__(self, buf: torch.Tensor, B: int, D: int, H: int, W: int, C: int, cs: int, bounded: bool = False) -> None:
        self.buf, self.B, self.D, self.H, self.W, self.C, self.cs = buf, B, D, H, W, C, cs
        self.bounded = bounded  # (engine.Act.bounded: the magnitudes do not scale with the sampler's state -- what the f16x2 kernels take)

    def _view(self, a: Act) -> Act:
        a.bounded = self.bounded
        return a

    def planes(self, b: int, d0: int, d1: int) -> Act:
        n = self.H * self.W * self.cs
        return self._view(Act(self.buf[(b * self.D + d0) * n : (b * self.D + d1) * n], d1 - d0, self.H, self.W, self.C, self.cs, True))

    def image(self) -> Act:
        r"""The volume as B images of (D H) x W pixels: what the norms and elementwise passes see."""
        return self._view(Act(self.buf, self.B, self.D * self.H, self.W, self.C, self.cs, True))

    def all_planes(self) -> Act:
        r"""The volume as B D images of H x W pixels: what one depth-tap launch over all planes sees."""
        return self._view(Act(self.buf, self.B * self.D, self.H, self.W, self.C, self.cs, True))


def new_vol(bld: Builder, B: int, D: int, H: int, W: int, C: int) -> Vol:
    a = bld.new_act(B * D, H, W, C)
    return Vol(a.buf, B, D, H, W, C, a.cs)


def free_vol(bld: Builder, v: Vol) -> None:
    bld.pool.release(v.buf)


def _depth_index(i: int, n: int, periodic: bool) -> int:
    r"""Plane index of a depth tap: itself, wrapped (circular padding) or -1 (zero padding)."""
    if 0 <= i < n:
        return i
    return i % n if periodic else -1


def conv3d(bld: Builder, x: Vol, conv, *, stride=1, periodic: bool = False, x1: Vol | None = None, up1=0,
           like: Vol | None = None, silu: bool = False, gate=None, gate_off: int = 0, gate_bstride: int = 0,
           res: Vol | None = None) -> Vol:
    r"""``conv``: holder of a (Cout, Cin, kd, kh, kw) weight (+ bias).  ``x1`` (read through nearest x 2^up1 upsampling,
    narrowed to ``like``'s size) is concatenated behind ``x``'s channels.  ``silu``: activation of the sum;
    ``gate`` / ``res``: out = res + gate * (sum + bias)."""
    w, bias = conv.weight, conv.bias
    cout, _, kd, kh, kw = w.shape
    assert kh == kw or True
    p = kd // 2
    Din, Hin, Win = (like.D, like.H, like.W) if like is not None else (x.D, x.H, x.W)
    sd_, sh_, sw_ = (stride, stride, stride) if isinstance(stride, int) else stride  # per-axis strides
    ud_, uh_, uw_ = (up1, up1, up1) if isinstance(up1, int) else up1  # per-axis log2 upsampling of x1
    Do = (Din + 2 * p - kd) // sd_ + 1
    Ho = (Hin + 2 * (kh // 2) - kh) // sh_ + 1
    Wo = (Win + 2 * (kw // 2) - kw) // sw_ + 1
    out = new_vol(bld, x.B, Do, Ho, Wo, cout)
    # (bounded sources and no residual of the stream: the sum of the taps is bounded by the weights, like a 2-D convolution's output)
    out.bounded = x.bounded and (x1 is None or x1.bounded) and (res is None or res.bounded)
    taps = [p] + [j for j in range(kd) if j != p]  # centre first: it exists for every output plane and writes it
    packs = {j: bld.pack_conv(w[:, :, j], bias if j == p else None, cin0=x.C if x1 is not None else None) for j in taps}
    # ---- ONE launch per depth tap for all planes of all samples (AzConvArgs.depth: the kernels' loaders take a plane whose
    # tap leaves its volume as zeros; the taps accumulate in place through `res`).  A stride-2 depth axis computes every plane
    # and keeps the even ones (twice the work of a layer that is a few per cent of the network, for 3 launches instead of
    # 3 Do B); a source read through depth upsampling (the decoder's merge convolutions) gets its planes duplicated first.
    fast = (cout > 4 and (gate is None or gate_bstride == 0 or x.B == 1)
            and (sd_ == 1 or (gate is None and res is None and not silu))
            and (x1 is None or (ud_ == 0 and x1.D == Din) or (ud_ == 1 and (Din + 1) // 2 <= x1.D)))
    if fast:
        g = dict(gate=gate, gate_off=gate_off, gate_bstride=0) if gate is not None else {}
        planes = lambda v: v.all_planes()  # noqa: E731
        x1u, kw_ = None, {}
        if x1 is not None:
            src1 = x1
            if ud_ == 1:  # nearest x2 along the depth axis (narrowed to Din planes): plane d of the wide volume = plane d >> 1
                x1u = new_vol(bld, x1.B, Din, x1.H, x1.W, x1.C)
                x1u.bounded = x1.bounded
                n = x1.H * x1.W * x1.cs
                if 2 * x1.D == Din:  # one launch per parity over all samples
                    for half_ in (0, 1):
                        bld.tape.add("az_token_copy_f32", x1u.buf.data_ptr(), 2, half_, x1.buf.data_ptr(), 1, 0, 1, x1.B * x1.D, n)
                else:  # odd depth (the reference narrows the upsampled volume, unet.py:253-255): per sample
                    for b_ in range(x1.B):
                        for half_ in (0, 1):
                            cnt = (Din - half_ + 1) // 2
                            if cnt > 0:
                                bld.tape.add("az_token_copy_f32", x1u.buf.data_ptr() + 4 * b_ * Din * n, 2, half_,
                                             x1.buf.data_ptr() + 4 * b_ * x1.D * n, 1, 0, 1, cnt, n)
                src1 = x1u
            kw_ = dict(src1=planes(src1), up1=(uh_, uw_), hin=Hin, win=Win)
        full = out if sd_ == 1 else new_vol(bld, x.B, Din, Ho, Wo, cout)  # (a strided depth axis: every plane, then every sd-th kept)
        full.bounded = out.bounded
        allo = planes(full)
        allr = planes(res) if res is not None else None
        # (the activation of the sum rides on the last tap's store: act 6 = silu(tap + what the earlier taps left; no pass of its own)
        fold = silu and len(taps) > 1 and sd_ == 1 and gate is None
        for j in taps:
            bld.conv(planes(x), packs[j], cout, stride=(sh_, sw_), out=allo, res=allr if j == p else allo, depth=(Din, j - p, periodic),
                     periodic=periodic, act=6 if (fold and j == taps[-1]) else 0, **g, **kw_)
        if sd_ > 1:  # out[d] = full[sd d]
            n = Ho * Wo * out.cs
            if Din % sd_ == 0:
                bld.tape.add("az_token_copy_f32", out.buf.data_ptr(), 1, 0, full.buf.data_ptr(), sd_, 0, 1, x.B * Do, n)
            else:  # the samples' planes do not group up across the batch: per sample, the last (partial) group apart
                for b_ in range(x.B):
                    whole = Din // sd_
                    if whole:
                        bld.tape.add("az_token_copy_f32", out.buf.data_ptr() + 4 * b_ * Do * n, 1, 0,
                                     full.buf.data_ptr() + 4 * b_ * Din * n, sd_, 0, 1, whole, n)
                    if Do > whole:  # plane sd * whole exists (Din % sd != 0) and is the last output plane
                        bld.tape.add("az_token_copy_f32", out.buf.data_ptr() + 4 * (b_ * Do + whole) * n, 1, 0,
                                     full.buf.data_ptr() + 4 * (b_ * Din + sd_ * whole) * n, 1, 0, 1, 1, n)
            free_vol(bld, full)
        if x1u is not None:
            free_vol(bld, x1u)
        if silu and not fold:
            bld.tape.add("az_silu_f32", out.buf.data_ptr(), out.buf.data_ptr(), out.buf.numel())
        return out
    for b in range(x.B):
        g = dict(gate=gate, gate_off=gate_off + b * gate_bstride, gate_bstride=0) if gate is not None else {}
        if (sd_, sh_, sw_) == (1, 1, 1) and x1 is None:  # contiguous plane ranges: one launch per (sample, tap[, wrap piece])
            for j in taps:
                o = j - p
                lo, hi = max(0, -o), min(Do, Din - o)  # output planes whose tap lies inside the volume
                pieces = [(lo, hi, lo + o)] if hi > lo else []
                if periodic and o != 0 and Din > 0:
                    if o < 0:  # planes [0, lo) read the last -o planes ... one plane at a time keeps the index arithmetic plain
                        pieces += [(d, d + 1, (d + o) % Din) for d in range(0, min(lo, Do))]
                    else:
                        pieces += [(d, d + 1, (d + o) % Din) for d in range(max(hi, 0), Do)]
                for d0, d1, s0 in pieces:
                    dst = out.planes(b, d0, d1)
                    first = j == p
                    bld.conv(x.planes(b, s0, s0 + (d1 - d0)), packs[j], cout, periodic=periodic, out=dst,
                             res=(res.planes(b, d0, d1) if res is not None else None) if first else dst, **g)
        else:
            for d in range(Do):
                for j in taps:
                    i = _depth_index(sd_ * d + j - p, Din, periodic)
                    if i < 0:
                        continue
                    dst = out.planes(b, d, d + 1)
                    first = j == p
                    kw_ = dict(src1=x1.planes(b, i >> ud_, (i >> ud_) + 1), up1=(uh_, uw_), hin=Hin, win=Win) if x1 is not None else {}
                    bld.conv(x.planes(b, i, i + 1), packs[j], cout, stride=(sh_, sw_), periodic=periodic, out=dst,
                             res=(res.planes(b, d, d + 1) if res is not None else None) if first else dst, **g, **kw_)
    if silu:
        bld.tape.add("az_silu_f32", out.buf.data_ptr(), out.buf.data_ptr(), out.buf.numel())
    return out


def upsample3d_nearest(bld: Builder, x: Vol, factors, like: Vol) -> Vol:
    r"""``narrow(Upsample(scale_factor=factors, mode="nearest")(x), like's size)`` for factors that are not powers of two:
    in-plane by ``az_upsample_nearest_f32`` on every plane, along the depth axis by a gather of whole planes with ATen's
    source index ``min(floor(d * float32(1 / s)), D - 1)`` (``az_gather_rows_f32``, index table built on the host)."""
    sd_, sh_, sw_ = factors
    planes = x.all_planes()
    wide = bld.upsample_nearest(planes, sh_, sw_, like.H, like.W) if (sh_, sw_) != (1, 1) or (x.H, x.W) != (like.H, like.W) else planes
    if sd_ == 1 and x.D == like.D:
        return Vol(wide.buf, x.B, x.D, like.H, like.W, x.C, x.cs, x.bounded)
    inv = torch.tensor(1.0 / sd_, dtype=torch.float32)
    src = torch.clamp(torch.floor(torch.arange(like.D, dtype=torch.float32) * inv).to(torch.int64), max=x.D - 1)
    idx = (torch.arange(x.B)[:, None] * x.D + src[None, :]).reshape(-1)
    out = new_vol(bld, x.B, like.D, like.H, like.W, x.C)
    out.bounded = x.bounded
    n = like.H * like.W * x.cs
    idx_dev = idx.to(bld.device)
    bld.tape.add("az_gather_rows_f32", out.buf.data_ptr(), wide.buf.data_ptr(), idx_dev.data_ptr(), x.B * like.D, n, x.B * x.D, keep=[idx_dev])
    if wide is not planes:
        bld.free(wide)
    return out


def block3d(blk, bld: Builder, x: Vol, D_mod: int, mod_rows: int, mod_jobs: list, keep_input: bool = False) -> Vol:
    r"""UNetBlock on a volume (reference ``unet.py:85-95``)."""
    Cc, cs = blk.channels, pad4(blk.channels)
    abc, bstride = ada_zero_triple(bld, blk.ada_zero, Cc, D_mod, mod_rows, mod_jobs)
    xi = x.image()
    if blk.norm_kind == "group":
        n_ = bld.group_norm(xi, blk.groups, scale=abc, shift=abc, scale_off=0, shift_off=cs, bstride=bstride)
    else:
        n_ = bld.row_norm(xi, 0 if blk.norm_kind == "layer" else 1, scale=abc, shift=abc, scale_off=0, shift_off=cs, bstride=bstride)
    nv = Vol(n_.buf, x.B, x.D, x.H, x.W, Cc, cs, n_.bounded)  # (the normalised volume: what the block's first convolution reads)
    c0, c3 = blk.ffn[0], blk.ffn[3]
    h1 = conv3d(bld, nv, c0, periodic=blk.periodic, silu=True)
    bld.free(n_)
    y = conv3d(bld, h1, c3, periodic=blk.periodic, gate=abc, gate_off=2 * cs, gate_bstride=bstride, res=x)
    free_vol(bld, h1)
    if not keep_input:
        free_vol(bld, x)
    return y


class UNet3DPlan:
    r"""Compiled forward for one (batch, D, H, W, modulation-rows) signature."""

    def __init__(self, net, B: int, D: int, H: int, W: int, mod_rows: int, device: torch.device) -> None:
        bld = self.bld = Builder(device, half=next(net.parameters()).dtype)
        cin = net.in_channels + net.cond_channels
        Dm = net.mod_features
        xa = Act(torch.empty(B * D * H * W * pad4(cin), dtype=torch.float32, device=device), B * D, H, W, cin, pad4(cin), True)
        self.x_in = Vol(xa.buf, B, D, H, W, cin, xa.cs)
        self.mod = torch.empty(max(mod_rows, 1), max(Dm, 1), dtype=torch.float32, device=device)
        self.out = torch.empty(B, net.out_channels, D, H, W, dtype=torch.float32, device=device)
        self.versions = net._param_versions()
        L = len(net.hid_blocks)
        stride, per = net.stride, net.periodic
        sv = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        pow2 = all(v in (1, 2, 4, 8, 16) for v in sv)
        up = tuple(v.bit_length() - 1 for v in sv) if pow2 else 0
        mod_jobs: list[tuple] = []
        cur = self.x_in
        skips: list[Vol] = []
        for i in range(L):
            first = net.descent[i][0]
            if i > 0:
                skips.append(cur)
            nxt = conv3d(bld, cur, first, stride=stride if i > 0 else 1, periodic=per)
            cur = nxt
            for j in range(net.hid_blocks[i]):
                cur = block3d(net.descent[i][1 + j], bld, cur, Dm, mod_rows, mod_jobs)
        for k in range(L):
            i = L - 1 - k
            mods = net.ascent[k]
            idx = 0
            if i + 1 < L:
                y = skips[i]
                if not pow2:  # Upsample(scale_factor=stride, "nearest") + narrow as tensors of their own (reference unet.py:186,250-255)
                    wide = upsample3d_nearest(bld, cur, sv, y)
                    free_vol(bld, cur)
                    cur = wide
                merged = conv3d(bld, y, mods[0], periodic=per, x1=cur, up1=up, like=y)
                free_vol(bld, cur)
                free_vol(bld, y)
                cur = merged
                idx = 1
            for j in range(net.hid_blocks[i]):
                cur = block3d(mods[idx + j], bld, cur, Dm, mod_rows, mod_jobs)
            idx += net.hid_blocks[i]
            if i == 0:
                head = conv3d(bld, cur, mods[idx], periodic=per)
                bld.tape.add("az_nhwc_to_nchw_f32", self.out.data_ptr(), head.buf.data_ptr(), B, net.out_channels, D * H * W, head.cs)
                free_vol(bld, cur)
        bld.finish()
        self.tape = bld.tape
        if mod_jobs:
            pre = mod_front_tape(bld, mod_jobs, self.mod, mod_rows, Dm)
            pre.extend(self.tape)
            self.tape = pre
