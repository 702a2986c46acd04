r"""ADM (guided-diffusion) UNet executed by gfx950 kernels.

Parameter layout and ``state_dict`` keys follow guided-diffusion's ``UNetModel`` exactly
(reference ``azula/plugins/adm/_src/unet.py:387-634``), so OpenAI checkpoints load
into ``denoiser.backbone`` unchanged.  The modules only hold parameters; the forward is a
compiled tape on the channel-padded NHWC layout:

* ResBlock (``_src/unet.py:227-247``): GN(32)+SiLU is a stats pass + one fused pass (with the
  2x2 average pool of ``down`` blocks folded in, ``:133``); nearest x2 of ``up`` blocks is a
  read-side shift of the conv gather and of the identity residual (``:104-106``); FiLM
  ``GN(h) * (1 + scale) + shift`` (``:239-243``) is folded into the second GN's scale/shift
  table; the skip (identity or 1x1 conv, ``:215``) is the conv epilogue's residual;
* decoder concat ``cat([h, hs.pop()])`` (``:631``) is never materialised: GroupNorm statistics,
  the normalise pass and the 1x1 skip conv all read the two sources in place;
* AttentionBlock (``:289-296``): GN -> 1x1 QKV GEMM -> flash attention (legacy ``(H 3 C)`` or new
  ``(3 H C)`` order, scale folded) -> zero-init 1x1 ``proj_out`` + residual epilogue;
* the sinusoidal timestep embedding (``_src/nn.py:90-108``) is a host table indexed on the
  device by the step's ``time_index``;
* ``dims=1`` (``conv_nd`` / ``avg_pool_nd``, ``_src/nn.py:50-77``): a (B, C, L) signal is a one-row image -- a 3-tap
  ``Conv1d`` filter is the middle row of a 3x3 one (the rows above and below only ever meet the zero padding), stride 2 /
  nearest x2 / the average pool act along the width alone (the anisotropic descriptor, pooling mode 2).
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib, engine
from ...engine import Act, Builder, pad8

__all__ = ["UNetModel"]


_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}  # conv_nd, reference _src/nn.py:50-61


def _zero(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


class ResBlock(nn.Module):
    r"""Parameter holder: in_layers (GN, SiLU, conv), emb_layers (SiLU, Linear), out_layers
    (GN, SiLU, Dropout, zero-init conv), skip_connection (reference ``_src/unet.py:140-225``)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, up=False, down=False,
                 dims=2):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.up, self.down, self.use_scale_shift_norm = up, down, use_scale_shift_norm
        oc = self.out_channels
        conv = _CONV[dims]
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), conv(channels, oc, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * oc if use_scale_shift_norm else oc))
        self.out_layers = nn.Sequential(
            nn.GroupNorm(32, oc), nn.SiLU(), nn.Dropout(p=dropout), _zero(conv(oc, oc, 3, padding=1))
        )
        self.skip_connection = nn.Identity() if oc == channels else conv(channels, oc, 1)


class Downsample(nn.Module):
    r"""Parameter holder of the ``resblock_updown=False`` downsampling layer: a stride-2 3x3 convolution ``op`` or a
    2x2 average pool (reference ``_src/unet.py:111-137``)."""

    def __init__(self, channels, use_conv, out_channels=None, dims=2):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.op = _CONV[dims](channels, self.out_channels, 3, stride=2 if dims != 3 else (1, 2, 2), padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool1d(2, 2) if dims == 1 else nn.AvgPool2d(2, 2) if dims == 2 else nn.AvgPool3d((1, 2, 2), (1, 2, 2))


class Upsample(nn.Module):
    r"""Parameter holder of the ``resblock_updown=False`` upsampling layer: nearest x2, then an optional 3x3
    convolution ``conv`` (reference ``_src/unet.py:82-109``)."""

    def __init__(self, channels, use_conv, out_channels=None, dims=2):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.conv = _CONV[dims](channels, self.out_channels, 3, padding=1)


class AttentionBlock(nn.Module):
    r"""Parameter holder: norm, qkv (Conv1d k=1), zero-init proj_out (reference ``_src/unet.py:250-296``)."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        assert channels % self.num_heads == 0
        self.new_order = use_new_attention_order
        self.norm = nn.GroupNorm(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = _zero(nn.Conv1d(channels, channels, 1))


class TimestepEmbedSequential(nn.Sequential):
    pass


def timestep_embedding_table(steps: int, dim: int, max_period: int = 10000) -> Tensor:
    r"""Rows = embeddings of the integer timesteps 0..steps-1, host fp32, reference op order
    (``_src/nn.py:90-108``): cos block || sin block."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = torch.arange(steps)[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class ADMPlan:
    r"""Compiled forward for (batch, H, W, embedding rows).  ``emb_rows`` = 1 when all samples
    share the timestep and there are no labels, else the batch size."""

    def __init__(self, net: "UNetModel", B: int, H: int, W: int, emb_rows: int, device, x_in: Act | None = None,
                 coef_ptr: int | None = None, frac: bool = False, D: int = 1) -> None:
        # a network cast to half precision keeps its activations in HBM in its own type (engine.HALF_ACT) when every kernel on its tape
        # has the typed form: 2-D / 1-D data, channel counts whose GroupNorm(32) groups are whole 4-channel chunks (C % 128 == 0: every
        # card of the plugin), a shared fp32 input (the second program of classifier-free guidance) with the typed plans' channel stride
        chans = sorted({m.num_channels for m in net.modules() if isinstance(m, nn.GroupNorm)})
        half_act = (net.dims != 3 and all(c % 128 == 0 for c in chans) and net.model_channels % 8 == 0
                    and (x_in is None or x_in.cs == pad8(net.in_channels)))  # (a shared input: the first program's, same stride)
        bld = self.bld = Builder(device, half=next(net.parameters()).dtype, half_act=half_act)
        mc, E = net.model_channels, 4 * net.model_channels
        self.versions = net._param_versions()
        self.emb_rows = emb_rows
        cin = net.in_channels
        one_d = net.dims == 1  # (B, C, L) signals: H = 1, every resampling acts along the width alone
        assert not one_d or H == 1
        # dims = 3: a (B, C, D, H, W) volume is B * D channel-padded NHWC planes -- every `Act` below has batch PB = B * D.  The
        # depth axis is never resampled (_src/unet.py:103-104,128: Upsample / Downsample act on the inner two axes), so a
        # Conv3d is the sum over its depth taps of 2-D convolutions of depth-shifted planes (AzConvArgs.depth, one launch per
        # tap accumulating through the residual operand, as nn/unet3d.py), the norms and elementwise passes see a sample as
        # one (D H) x W image (`vol`), the attention blocks as D H W tokens.
        three_d = net.dims == 3
        assert three_d or D == 1
        PB = B * D
        up2 = (0, 1) if one_d else 1  # log2 of the nearest upsampling per axis
        down2 = (1, 2) if one_d else 2  # stride per axis
        pool2 = 2 if one_d else 1  # az_affine_act_f32's pooling mode: 1x2 or 2x2
        wino = False if one_d else None  # (a one-row map: the direct kernel; F(2x2, 3x3) would compute a second, discarded row)

        def packed(conv, **kw):
            r"""Filter of a ConvNd layer; a 3-tap Conv1d becomes the middle row of a 3x3 filter."""
            w = conv.weight
            if w.ndim == 3 and w.shape[-1] == 3:
                w2 = torch.zeros(*w.shape[:2], 3, 3, dtype=w.dtype, device=w.device)
                w2[:, :, 1, :] = w.detach()
                w = w2
            return bld.pack_conv(w, conv.bias, **kw)

        def halved(a: Act) -> tuple[int, int]:
            return (a.H, a.W // 2) if one_d else (a.H // 2, a.W // 2)

        def vol(a: Act | None) -> Act | None:
            r"""The B samples of a plane stack as (D H) x W images (same memory); the producing convolution's GroupNorm
            partials, one set per plane, are D sets per sample."""
            if a is None or not three_d:
                return a
            v = Act(a.buf, B, D * a.H, a.W, a.C, a.cs, True)
            v.bounded = a.bounded
            if a.gn_quads is not None:
                v.gn_quads = (a.gn_quads[0], D * a.gn_quads[1])
            return v

        def planes(a: Act) -> Act:
            r"""Inverse of `vol` for a pass's result (pooled or not): B (D H') x W' images -> B D planes of H' x W'."""
            if not three_d:
                return a
            p_ = Act(a.buf, PB, a.H // D, a.W, a.C, a.cs, a.pinned)
            p_.bounded = a.bounded
            return p_

        def group_norm(x: Act, *args, x1: Act | None = None, **kw) -> Act:
            return planes(bld.group_norm(vol(x), *args, x1=vol(x1), **kw))

        def conv(src: Act, layer, cout: int, *, cin0=None, **kw) -> Act | None:
            r"""conv_nd layer on `src` (| kw["src1"] concatenated).  Conv3d with three depth taps: the centre tap first (it exists for
            every plane; bias, residual), the others accumulate in place; the moments for a following GroupNorm come from the last."""
            w = layer.weight
            if not (three_d and w.ndim == 5 and w.shape[2] > 1):
                if w.ndim == 5:  # 1 x 1 x 1
                    return bld.conv(src, bld.pack_conv(w[:, :, 0], layer.bias, cin0=cin0), cout, **kw)
                return bld.conv(src, packed(layer, cin0=cin0), cout, **kw)
            assert w.shape[2] == 3
            gn_stats, dst = kw.pop("gn_stats", False), kw.pop("dst_nchw", None)
            assert dst is None
            res, res_up = kw.pop("res", None), kw.pop("res_up", 0)
            # a depth tap that reads only padding (|j - 1| >= D: a single-plane volume) contributes zero and is not launched; the
            # moments ride on the LAST tap that is
            taps = [j for j in (0, 2) if abs(j - 1) < D]
            out = bld.conv(src, bld.pack_conv(w[:, :, 1], layer.bias, cin0=cin0), cout, res=res, res_up=res_up, depth=(D, 0),
                           gn_stats=gn_stats and not taps, **kw)
            for j in taps:
                bld.conv(src, bld.pack_conv(w[:, :, j], None, cin0=cin0), cout, res=out, out=out, depth=(D, j - 1),
                         gn_stats=gn_stats and j == taps[-1], **kw)
            return out

        c_first = net.input_blocks[0][0]
        # the first convolution reads the latent PLANAR (x_in = the loop's own (B, C, H, W) layout, channel stride 0): Builder.conv_stem
        self.planar = (engine.STEM_PLANAR and cin <= 4 and not three_d and isinstance(c_first, (nn.Conv1d, nn.Conv2d)) and c_first.weight.shape[-1] == 3
                       and c_first.out_channels % 4 == 0 and bld.half is None and (x_in is None or x_in.cs == 0))
        if x_in is not None:
            self.x_in = x_in
        elif self.planar:
            self.x_in = Act(torch.empty(B * cin * H * W, dtype=torch.float32, device=device), B, H, W, cin, 0, True)
        else:
            self.x_in = Act(torch.zeros(PB * H * W * bld.pad(cin), dtype=torch.float32, device=device), PB, H, W, cin, bld.pad(cin), True)
        self.out = torch.empty((B, net.out_channels) + ((D,) if three_d else ()) + (H, W), dtype=torch.float32, device=device)
        self.table = bld.const(timestep_embedding_table(net.table_steps, mc))
        self.t_idx = torch.zeros(emb_rows, dtype=torch.int64, device=device)
        self.labels = torch.zeros(B, dtype=torch.int64, device=device) if net.num_classes is not None else None
        temb = bld.empty(emb_rows, mc)
        row0 = torch.zeros(B, dtype=torch.int64, device=device)  # index vector that replicates a single row per sample
        bld.tape.keep.append(row0)  # (int64: Builder.const would cast it to fp32 and halve its size)
        if coef_ptr is not None:  # fused sampling: row index = the step's time_index, read on the device
            assert emb_rows == 1 or net.num_classes is not None
            bld.tape.add("az_gather_step_row_f32", temb.data_ptr(), self.table.data_ptr(), coef_ptr, 0, mc, net.table_steps)
            trow = 1
        elif frac:  # fractional timesteps (_src/nn.py:90-108 "these may be fractional"): the sinusoid on the device, no table
            assert mc % 2 == 0, "odd model_channels with fractional timesteps"
            self.t_frac = torch.zeros(emb_rows, dtype=torch.float32, device=device)
            bld.tape.add("az_timestep_embedding_f32", temb.data_ptr(), mc, self.t_frac.data_ptr(), 1, emb_rows, mc // 2, 10000.0)
            trow = emb_rows
        else:
            bld.tape.add("az_gather_rows_f32", temb.data_ptr(), self.table.data_ptr(), self.t_idx.data_ptr(), emb_rows, mc, net.table_steps)
            trow = emb_rows
        te0, te2 = net.time_embed[0], net.time_embed[2]
        hid = bld.empty(trow, E)
        emb = bld.empty(emb_rows, E)
        bld.linear_small(hid, E, temb, mc, bld.const(te0.weight), bld.const(te0.bias), trow, E, mc, 0, 1)
        if net.num_classes is None:
            bld.linear_small(emb, E, hid, E, bld.const(te2.weight), bld.const(te2.bias), trow, E, E, 0, 0)
        else:  # emb = time_embed(t) + label_emb(y), per sample (_src/unet.py:621-623)
            tbase = bld.empty(trow, E)
            bld.linear_small(tbase, E, hid, E, bld.const(te2.weight), bld.const(te2.bias), trow, E, E, 0, 0)
            lab = bld.empty(B, E)
            lw = bld.const(net.label_emb.weight)
            bld.tape.add("az_gather_rows_f32", lab.data_ptr(), lw.data_ptr(), self.labels.data_ptr(), B, E, net.num_classes)
            ones = bld.const(torch.ones(1))
            if trow == 1:
                tb = bld.empty(B, E)
                bld.tape.add("az_gather_rows_f32", tb.data_ptr(), tbase.data_ptr(), row0.data_ptr(), B, E, 1)
                tbase = tb
            bld.tape.add("az_axpby_f32", emb.data_ptr(), ones.data_ptr(), tbase.data_ptr(), ones.data_ptr(), lab.data_ptr(), 1, B * E, 0)
        ebs = E if emb_rows > 1 else 0  # batch stride of emb rows
        film_at = len(bld.tape.ops)  # every ResBlock's FiLM projection reads only `emb`: one grouped launch here
        film_jobs: list[tuple] = []

        def resblock(rb: ResBlock, x: Act, x1: Act | None = None) -> Act:
            r"""x (| x1 concatenated) -> block output.  Does not free its inputs."""
            oc, ocs = rb.out_channels, bld.pad(rb.out_channels)
            gi, ci = rb.in_layers[0], rb.in_layers[2]
            go, co = rb.out_layers[0], rb.out_layers[3]
            # FiLM table: emb_layers = SiLU -> Linear(E, 2*oc); (scale | shift) padded to ocs each
            lin = rb.emb_layers[1]
            nf = 2 if rb.use_scale_shift_norm else 1  # (scale | shift), or the additive embedding alone
            w = torch.zeros(nf * ocs, E, dtype=torch.float32, device=device)
            b_ = torch.zeros(nf * ocs, dtype=torch.float32, device=device)
            for n in range(nf):
                w[n * ocs : n * ocs + oc] = lin.weight.detach()[n * oc : (n + 1) * oc]
                b_[n * ocs : n * ocs + oc] = lin.bias.detach()[n * oc : (n + 1) * oc]
            film = bld.empty(emb_rows, nf * ocs)
            film_jobs.append((film, bld.const(w), bld.const(b_), nf * ocs))
            fbs = nf * ocs if emb_rows > 1 else 0
            # h = conv(updown(SiLU(GN(x))))
            n1 = group_norm(x, 32, weight=bld.const(gi.weight), bias=bld.const(gi.bias), act=1, pool=pool2 if rb.down else 0, x1=x1)
            h = conv(n1, ci, oc, up0=up2 if rb.up else 0, winograd=wino,
                     gn_stats=rb.use_scale_shift_norm)  # -> out_layers' GroupNorm
            bld.free(n1)
            if rb.use_scale_shift_norm:  # h = SiLU(GN(h) * (1 + scale) + shift)
                n2 = group_norm(h, 32, weight=bld.const(go.weight), bias=bld.const(go.bias), scale=film, shift=film,
                                scale_off=0, shift_off=ocs, bstride=fbs, act=1)
            else:  # h = SiLU(GN(h + emb_out)), _src/unet.py:244-246: the statistics are those of the SUM -> one more pass
                eb = film
                if emb_rows == 1 and B > 1:  # the pass wants one row per sample
                    eb = bld.empty(B, ocs)
                    bld.tape.add("az_gather_rows_f32", eb.data_ptr(), film.data_ptr(),
                                 row0.data_ptr(), B, ocs, 1)
                he = bld.new_act(PB, h.H, h.W, oc)
                bld._affine_act(he, h, None, 0, bld.const(torch.ones(B * ocs)).data_ptr(), eb.data_ptr(), B, D * h.H, h.W, ocs, 0, 0)
                n2 = group_norm(he, 32, weight=bld.const(go.weight), bias=bld.const(go.bias), act=1)
                bld.free(he)
            bld.free(h)
            # skip path
            if rb.down:
                assert x1 is None
                ones, zeros = bld.const(torch.ones(B * x.cs)), bld.const(torch.zeros(B * x.cs))
                xs = bld.new_act(PB, *halved(x), x.C)
                bld._affine_act(xs, x, None, 0, ones.data_ptr(), zeros.data_ptr(), B, D * x.H, x.W, x.cs, 0, pool2)
                out = conv(n2, co, oc, res=xs, gn_stats=True, winograd=wino)
                bld.free(xs)
            elif rb.up:
                assert x1 is None
                out = conv(n2, co, oc, res=x, res_up=1, gn_stats=True, winograd=wino)  # (one row: oh >> 1 = 0)
            elif isinstance(rb.skip_connection, nn.Identity):
                assert x1 is None
                out = conv(n2, co, oc, res=x, gn_stats=True, winograd=wino)
            else:
                sc = rb.skip_connection
                skip = conv(x, sc, oc, cin0=x.C if x1 is not None else None, src1=x1)
                out = conv(n2, co, oc, res=skip, gn_stats=True, winograd=wino)
                bld.free(skip)
            bld.free(n2)
            return out

        def attention(ab: AttentionBlock, x: Act) -> Act:
            Cc = ab.channels
            n_ = group_norm(x, 32, weight=bld.const(ab.norm.weight), bias=bld.const(ab.norm.bias))
            tok = Act(n_.buf, B, D * n_.H * n_.W, 1, Cc, n_.cs, True)
            tok.bounded = n_.bounded  # (the normalised tokens: the same memory)
            ch = Cc // ab.num_heads
            chp = engine.attn_padded_dim(ch, bld.half)
            order = "3HC" if ab.new_order else "H3C"
            xt = Act(x.buf, B, D * x.H * x.W, 1, Cc, x.cs, True)
            if chp != ch:  # a head size the kernels are not instantiated for: zero-padded heads (engine.ATTN_HEAD_DIMS; G24)
                wq, bq = engine.pad_qkv_heads(ab.qkv.weight, ab.qkv.bias, ab.num_heads, ch, chp, order)
                qkv = bld.conv(tok, bld.pack_conv(wq, bq), 3 * ab.num_heads * chp)
                att = bld.attention(qkv, ab.num_heads, order, False, 1.0 / math.sqrt(ch), norm_dim=ch)
                bld.free(qkv)
                o = bld.conv(att, bld.pack_conv(engine.pad_proj_heads(ab.proj_out.weight, ab.num_heads, ch, chp), ab.proj_out.bias), Cc, res=xt)
            else:
                qkv = bld.conv(tok, bld.pack_conv(ab.qkv.weight, ab.qkv.bias), 3 * Cc)
                att = bld.attention(qkv, ab.num_heads, order, False, 1.0 / math.sqrt(ch))
                bld.free(qkv)
                o = bld.conv(att, bld.pack_conv(ab.proj_out.weight, ab.proj_out.bias), Cc, res=xt)
            bld.free(att)
            bld.free(n_)
            return Act(o.buf, PB, x.H, x.W, Cc, o.cs)

        def run(block: nn.Sequential, h: Act, h1: Act | None = None) -> Act:
            for layer in block:
                if isinstance(layer, (nn.Conv1d, nn.Conv2d)) and h is self.x_in and self.planar:
                    nh = bld.conv_stem(h.buf, B, cin, H, W, packed(layer), layer.out_channels, gn_stats=True)
                elif isinstance(layer, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
                    nh = conv(h, layer, layer.out_channels, gn_stats=True, winograd=wino)
                elif isinstance(layer, ResBlock):
                    nh = resblock(layer, h, h1)
                elif isinstance(layer, Downsample):
                    if layer.use_conv:
                        nh = conv(h, layer.op, layer.out_channels, stride=down2)
                    else:  # AvgPoolNd(2, 2): the pooling form of the elementwise pass with S = 1, T = 0
                        nh = bld.new_act(PB, *halved(h), h.C)
                        bld._affine_act(nh, h, None, 0, bld.const(torch.ones(B * h.cs)).data_ptr(),
                                        bld.const(torch.zeros(B * h.cs)).data_ptr(), B, D * h.H, h.W, h.cs, 0, pool2)
                elif isinstance(layer, Upsample):
                    if layer.use_conv:  # nearest x2 is a read-side shift of the conv gather
                        nh = conv(h, layer.conv, layer.out_channels, up0=up2, winograd=wino)
                    else:  # nearest x2 alone: the same gather under an identity 1x1 filter (exact: one product per output)
                        eye = torch.eye(h.C, dtype=torch.float32, device=device)
                        nh = bld.conv(h, bld.pack_conv(eye, None), h.C, up0=up2)
                else:
                    nh = attention(layer, h)
                if h1 is None and h not in hs and h is not self.x_in:
                    bld.free(h)
                h, h1 = nh, None
            return h

        hs: list[Act] = []
        h = self.x_in
        for block in net.input_blocks:
            h = run(block, h)
            hs.append(h)
        hs_keep = list(hs)
        h = run(net.middle_block, h)
        for block in net.output_blocks:
            skip = hs.pop()
            nh = run(block, h, skip)
            if h not in hs_keep:
                bld.free(h)
            bld.free(skip)
            h = nh
        go, co = net.out[0], net.out[2]
        n_ = group_norm(h, 32, weight=bld.const(go.weight), bias=bld.const(go.bias), act=1)
        if three_d:  # (the NCHW epilogue writes per-plane images: a volume's (C, D, H, W) order takes a pass of its own)
            head = conv(n_, co, net.out_channels, winograd=wino)
            bld.tape.add("az_nhwc_to_nchw_f32", self.out.data_ptr(), head.ptr, B, net.out_channels, D * H * W, head.cs)
        else:
            bld.conv(n_, packed(co), net.out_channels, dst_nchw=self.out, winograd=wino)
        bld.finish()
        if film_jobs:
            from ..._lib import AzLinearGroup, lib

            nj = len(film_jobs)
            groups = (AzLinearGroup * nj)()
            for i, (film, w, b_, n_out) in enumerate(film_jobs):
                g = groups[i]
                g.y, g.x, g.W, g.bias = film.data_ptr(), emb.data_ptr(), w.data_ptr(), b_.data_ptr()
                g.ldy, g.ldx, g.N, g.K = n_out, E, n_out, E
            gdev = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).to(device)
            bld.tape.keep.append(gdev)
            bld.tape.ops.insert(film_at, (
                lib().az_linear_small_grouped_f32, (gdev.data_ptr(), nj, max(j[3] for j in film_jobs), emb_rows, 1, 0),
                "az_linear_small_grouped_f32",
            ))
        self.tape = bld.tape


class UNetModel(nn.Module):
    r"""guided-diffusion ``UNetModel`` (reference ``_src/unet.py:387-634``), gfx950-native forward.

    ``dims=2`` (all of the plugin's cards), ``dims=1`` ((B, C, L) signals, run as one-row images) and ``dims=3`` ((B, C, D, H, W)
    volumes as B D planes: a Conv3d is three depth-tap launches of the 2-D kernels, G23).  The cards use
    ``resblock_updown=True, use_scale_shift_norm=True``; guided-diffusion's defaults (``h + emb`` instead of FiLM,
    ``Downsample`` / ``Upsample`` layers with or without ``conv_resample``) are built too (G14).
    """

    def __init__(
        self,
        image_size,
        in_channels,
        model_channels,
        out_channels,
        num_res_blocks,
        attention_resolutions,
        dropout=0,
        channel_mult=(1, 2, 4, 8),
        conv_resample=True,
        dims=2,
        num_classes=None,
        use_checkpoint=False,
        num_heads=1,
        num_head_channels=-1,
        num_heads_upsample=-1,
        use_scale_shift_norm=False,
        resblock_updown=False,
        use_new_attention_order=False,
    ) -> None:
        super().__init__()
        if dims not in (1, 2, 3):
            raise ValueError(f"unsupported dimensions: {dims}")
        self.dims = dims
        conv = _CONV[dims]
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_classes = out_channels, num_classes
        self.table_steps = 1000
        E = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, E), nn.SiLU(), nn.Linear(E, E))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, E)

        def res(ch, oc=None, **kw):
            return ResBlock(ch, E, dropout, out_channels=oc, use_scale_shift_norm=use_scale_shift_norm, dims=dims, **kw)

        def attn(ch, heads):
            return AttentionBlock(ch, num_heads=heads, num_head_channels=num_head_channels,
                                  use_new_attention_order=use_new_attention_order)

        ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv(in_channels, ch, 3, padding=1))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, out_channels=ch, dims=dims)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch), attn(ch, num_heads), res(ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [res(ch + ich, int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown else Upsample(ch, conv_resample, out_channels=ch, dims=dims))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), _zero(conv(ch, out_channels, 3, padding=1)))
        self._plans: dict = {}

    def _param_versions(self) -> tuple:
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def plan(self, B, H, W, emb_rows, device, x_in: Act | None = None, coef_ptr: int | None = None, tag=None, frac=False,
             D: int = 1) -> ADMPlan:
        key = (B, D, H, W, emb_rows, str(device), x_in.ptr if x_in is not None else None, coef_ptr, tag, frac)
        p = self._plans.get(key)
        if p is None or p.versions != self._param_versions():
            p = ADMPlan(self, B, H, W, emb_rows, device, x_in=x_in, coef_ptr=coef_ptr, frac=frac, D=D)
            self._plans[key] = p
        return p

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, timesteps: Tensor, y: Tensor | None = None) -> Tensor:
        r"""x: (N, C, H, W) [dims = 1: (N, C, L); dims = 3: (N, C, D, H, W)]; timesteps: (N,) or (1,) integer indices or fractional values; y: (N,) labels
        iff class-conditional."""
        assert (y is not None) == (self.num_classes is not None), (
            "must specify y if and only if the model is class-conditional"
        )
        from ...nn.utils import backbone_io_dtype

        out_dtype = backbone_io_dtype(self, x, "azula_amd ADM UNetModel")
        x = x.to(torch.float32).contiguous()
        assert x.ndim == self.dims + 2, f"dims={self.dims}: expected a {self.dims + 2}-d input, got {tuple(x.shape)}"
        shape = x.shape
        if self.dims == 1:
            x = x[:, :, None, :]
        D = x.shape[2] if self.dims == 3 else 1
        B, Cin, H, W = x.shape[0], x.shape[1], x.shape[-2], x.shape[-1]
        timesteps = timesteps.reshape(-1)
        frac = torch.is_floating_point(timesteps)  # (azula itself passes integer indices; guided-diffusion allows fractions)
        rows = B if (timesteps.numel() > 1 or self.num_classes is not None) else 1
        p = self.plan(B, H, W, rows, x.device, frac=frac, D=D)
        s = _lib.stream_ptr()
        if p.planar:
            p.x_in.buf.copy_(x.reshape(-1))
        else:
            _lib.call("az_nchw_to_nhwc_f32", p.x_in.ptr, x.data_ptr(), None, B, Cin, D * H * W, p.x_in.cs, s)
        if frac:
            p.t_frac.copy_(timesteps.to(torch.float32).expand(rows) if timesteps.numel() == 1 else timesteps.to(torch.float32))
        else:
            p.t_idx.copy_(timesteps.to(torch.int64).expand(rows) if timesteps.numel() == 1 else timesteps.to(torch.int64))
        if y is not None:
            assert y.shape == (B,)
            p.labels.copy_(y.to(torch.int64))
        p.tape.run(s)
        return p.out.reshape(B, self.out_channels, *shape[2:]).to(out_dtype, copy=True)
