r"""JiT ("Just image Transformer") backbone executed by gfx950 kernels.

Parameter holder with the ``state_dict`` keys of the reference's vendored model
(``azula/plugins/jit/_src/model.py:218-381``, from LTH14/JiT) plus a compiled forward built from the
same kernels as ``azula_amd.nn.vit``:

* bottleneck patch embedding = patchify remap + two MFMA GEMMs, the fixed sin-cos positional table
  added in the second GEMM's epilogue;
* conditioning c = MLP(sinusoid(t)) + label embedding: ``az_timestep_embedding_f32``, the small-M
  linear kernel, ``az_gather_rows_f32``; the 6-way adaLN projections of ALL blocks (and the final layer's) are
  one MFMA GEMM on SiLU(c), issued once per forward;
* block: weighted RMSNorm + modulate is one row pass (``az_rownorm_mod_f32``); the q/k RMSNorm gains,
  the 2-D rotary embedding and the 1/sqrt(d) scale are folded into the attention kernel's operand loads
  ('(3 H C)' fused-QKV layout read in place); ``x + gate * proj(.)`` and ``x + gate * w3(.)`` are GEMM
  epilogues; SwiGLU reads w12's output once (w12's rows are interleaved at build time so the kernel's
  pair layout applies);
* the in-context class tokens are prepended by two token-window kernels at ``in_context_start`` and dropped
  before the final layer; the final linear's rows are permuted at build time from the model's (p, q, c)
  feature order to the unpatchify kernel's (c, p, q).
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib, engine
from ...engine import Act, Builder

__all__ = ["JiT", "JiT_models"]


class _Gain(nn.Module):
    r"""RMSNorm gain holder (``_src/util.py:149-157``): key ``weight``."""

    def __init__(self, n: int) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))


class _Attention(nn.Module):
    def __init__(self, dim: int, heads: int) -> None:
        super().__init__()
        self.q_norm, self.k_norm = _Gain(dim // heads), _Gain(dim // heads)
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _SwiGLU(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        hidden = int(hidden * 2 / 3)  # _src/model.py:153
        self.w12 = nn.Linear(dim, 2 * hidden)
        self.w3 = nn.Linear(hidden, dim)


def _ada(dim: int, ways: int) -> nn.Sequential:
    seq = nn.Sequential(nn.SiLU(), nn.Linear(dim, ways * dim))
    nn.init.zeros_(seq[1].weight)
    nn.init.zeros_(seq[1].bias)
    return seq


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: float) -> None:
        super().__init__()
        self.norm1 = _Gain(dim)
        self.attn = _Attention(dim, heads)
        self.norm2 = _Gain(dim)
        self.mlp = _SwiGLU(dim, int(dim * mlp_ratio))
        self.adaLN_modulation = _ada(dim, 6)


class _Final(nn.Module):
    def __init__(self, dim: int, patch: int, channels: int) -> None:
        super().__init__()
        self.norm_final = _Gain(dim)
        self.linear = nn.Linear(dim, patch * patch * channels)
        self.adaLN_modulation = _ada(dim, 2)
        nn.init.zeros_(self.linear.weight)
        nn.init.zeros_(self.linear.bias)


def sincos_table(dim: int, grid: int) -> Tensor:
    r"""Fixed 2-D sin-cos positional table (grid^2, dim): the column index drives the first half of the
    features, the row index the second; each half is [sin | cos] over dim/4 log-spaced frequencies
    (fp64, reference ``_src/util.py:166-212``)."""
    omega = 1.0 / 10000 ** (torch.arange(dim // 4, dtype=torch.float64) / (dim / 4.0))
    idx = torch.arange(grid, dtype=torch.float64)
    ang = idx[:, None] * omega[None]
    enc = torch.cat((ang.sin(), ang.cos()), dim=1)  # (grid, dim / 2)
    cols = enc[None, :, :].expand(grid, grid, -1)
    rows = enc[:, None, :].expand(grid, grid, -1)
    return torch.cat((cols, rows), dim=-1).reshape(grid * grid, dim).float()


def rotary_tables(head_dim: int, heads: int, grid: int, ctx: int) -> tuple[Tensor, Tensor]:
    r"""cos / sin of the rotation angle per (token, head, channel pair), laid out as ``az_attention_f32``
    reads them: (ctx + grid^2, heads, head_dim / 2).  Pairs [0, d/4) turn with the patch row, [d/4, d/2)
    with the patch column, frequencies 10000^(-2j / (d/2)); context tokens are not rotated
    (reference ``_src/util.py:100-143`` with dim = head_dim / 2)."""
    quarter = head_dim // 4
    freqs = 1.0 / (10000.0 ** (torch.arange(0, head_dim // 2, 2)[:quarter].float() / (head_dim // 2)))
    pos = torch.arange(grid) / grid * grid
    ang = pos[:, None] * freqs[None]  # (grid, d/4)
    full = torch.cat((ang[:, None, :].expand(grid, grid, quarter), ang[None, :, :].expand(grid, grid, quarter)), dim=-1)
    full = full.reshape(grid * grid, 2 * quarter)
    cos, sin = full.cos(), full.sin()
    if ctx:
        cos = torch.cat((torch.ones(ctx, 2 * quarter), cos))
        sin = torch.cat((torch.zeros(ctx, 2 * quarter), sin))
    expand = lambda t: t[:, None, :].expand(-1, heads, -1).contiguous()  # noqa: E731
    return expand(cos), expand(sin)


class JiTPlan:
    r"""Compiled forward for one (batch, shared/per-sample time) signature."""

    def __init__(self, net: "JiT", B: int, t_shared: bool, device) -> None:
        # a network cast to half precision keeps its activations in HBM in its own type (engine.HALF_ACT): widths in multiples of 8,
        # SwiGLU through the GEMM's epilogue
        half_act = net.hidden_size % 8 == 0 and all((blk.mlp.w12.weight.shape[0] // 2) % 8 == 0 for blk in net.blocks) and net.hidden_size <= 4096
        bld = self.bld = Builder(device, half=net.pos_embed.dtype, half_act=half_act)
        Hd, heads, p, Z = net.hidden_size, net.num_heads, net.patch_size, net.in_channels
        S = net.input_size
        grid = S // p
        L, Lc = grid * grid, net.in_context_len
        hd = Hd // heads
        self.versions = net._param_versions()
        f32 = dict(dtype=torch.float32, device=device)
        self.x_nchw = torch.empty(B, Z, S, S, **f32)
        self.t = torch.zeros(1 if t_shared else B, **f32)
        self.labels = torch.zeros(B, dtype=torch.int64, device=device)
        self.out = torch.empty(B, Z, S, S, **f32)
        tape = bld.tape

        # ---- conditioning: c = t_embedder(t) + y_embedder(y)     (_src/model.py:353-355)
        F_ = net.t_embedder.frequency_embedding_size
        m0, m2 = net.t_embedder.mlp[0], net.t_embedder.mlp[2]
        table = net.y_embedder.embedding_table.weight
        freq, hid, t_emb, y_emb, c = (bld.empty(B, n) for n in (F_, Hd, Hd, Hd, Hd))
        ones = bld.const(torch.ones(1))
        tape.add("az_timestep_embedding_f32", freq.data_ptr(), F_, self.t.data_ptr(), 0 if t_shared else 1, B, F_ // 2, 10000.0)
        bld.linear_small(hid, Hd, freq, F_, bld.const(m0.weight), bld.const(m0.bias), B, Hd, F_, 0, 1)
        bld.linear_small(t_emb, Hd, hid, Hd, bld.const(m2.weight), bld.const(m2.bias), B, Hd, Hd, 0, 0)
        tape.add("az_gather_rows_f32", y_emb.data_ptr(), bld.const(table).data_ptr(), self.labels.data_ptr(), B, Hd, table.shape[0])
        tape.add("az_axpby_f32", c.data_ptr(), ones.data_ptr(), t_emb.data_ptr(), ones.data_ptr(), y_emb.data_ptr(), 1, B * Hd, 0)
        # every adaLN projection reads only SiLU(c): the 6-way projections of all blocks and the final layer's
        # 2-way one are ONE GEMM (B x Hd) @ (Hd x (6 depth + 2) Hd) on the MFMA kernel, issued once per forward
        heads_ = [blk.adaLN_modulation[1] for blk in net.blocks] + [net.final_layer.adaLN_modulation[1]]
        w_all = torch.cat([m.weight.detach() for m in heads_]).to(device)
        b_all = torch.cat([m.bias.detach() for m in heads_]).to(device)
        c_act = Act(bld.empty(B * Hd), 1, B, 1, Hd, Hd, True)
        tape.add("az_silu_f32", c_act.ptr, c.data_ptr(), B * Hd)
        mods = bld.conv(c_act, bld.pack_conv(w_all, b_all), w_all.shape[0], out_f32=True)  # (the modulation table stays fp32)
        mods.pinned = True
        mod, MS = mods.buf, mods.cs  # row stride of the modulation table

        # ---- bottleneck patch embedding + fixed positional table  (_src/model.py:16-43,358-359)
        e = net.x_embedder
        tokens = bld.new_act(B, L, 1, Z * p * p, pinned=True, f32=True)  # (the plan's input stays fp32)
        tape.add("az_patchify_f32", tokens.ptr, self.x_nchw.data_ptr(), None, B, Z, S, S, p, tokens.cs)
        low = bld.conv(tokens, bld.pack_conv(e.proj1.weight.detach().reshape(e.proj1.out_channels, -1), None), e.proj1.out_channels)
        pos_t = bld.const(net.pos_embed.detach().reshape(-1))
        if bld.half_act:  # (the residual operand has the destination's element type)
            pos_t = pos_t.to(bld.half)
            tape.keep.append(pos_t)
        pos = Act(pos_t, 1, L, 1, Hd, Hd, True)
        x = bld.conv(low, bld.pack_conv(e.proj2.weight.detach().reshape(Hd, -1), e.proj2.bias), Hd, res=_Shared(pos))
        bld.free(low)

        hdp = engine.attn_padded_dim(hd, bld.half)

        def rope_tables(ctx: int) -> tuple:
            cos, sin = rotary_tables(hd, heads, grid, ctx)  # (tokens, heads, hd / 2)
            if hdp != hd:  # padded pairs do not turn
                cos = torch.cat([cos, cos.new_ones(*cos.shape[:-1], (hdp - hd) // 2)], dim=-1).contiguous()
                sin = torch.cat([sin, sin.new_zeros(*sin.shape[:-1], (hdp - hd) // 2)], dim=-1).contiguous()
            return bld.const(cos), bld.const(sin)

        rope_img = rope_tables(0)
        rope_ctx = rope_tables(Lc) if Lc else rope_img
        for i, blk in enumerate(net.blocks):
            if Lc and i == net.in_context_start:  # prepend the class tokens (_src/model.py:364-367)
                wide = bld.new_act(B, L + Lc, 1, Hd)
                ctx_pos = bld.const(net.in_context_posemb.detach().reshape(-1))
                if wide.half:  # 2-byte tokens: typed fill; the copy moves Hd / 2 floats per token
                    tape.add("az_token_fill_h16", wide.ptr, L + Lc, 0, Lc, y_emb.data_ptr(), Hd, ctx_pos.data_ptr(), B, Hd, 2 if bld.half == torch.float16 else 1)
                    tape.add("az_token_copy_f32", wide.ptr, L + Lc, Lc, x.ptr, L, 0, L, B, Hd // 2)
                else:
                    tape.add("az_token_fill_f32", wide.ptr, L + Lc, 0, Lc, y_emb.data_ptr(), Hd, ctx_pos.data_ptr(), B, Hd)
                    tape.add("az_token_copy_f32", wide.ptr, L + Lc, Lc, x.ptr, L, 0, L, B, Hd)
                bld.free(x)
                x = wide
            m0_ = 6 * Hd * i  # this block's columns: shift_a | scale_a | gate_a | shift_m | scale_m | gate_m
            n1 = bld.row_norm(x, 1, weight=bld.const(blk.norm1.weight), scale=mod, shift=mod, scale_off=m0_ + Hd,
                              shift_off=m0_, bstride=MS, eps=1e-6)
            at = blk.attn
            rope_i = rope_img if i < net.in_context_start else rope_ctx
            gains = (bld.const(at.q_norm.weight), bld.const(at.k_norm.weight))
            # q / k RMS norm, gains and RoPE in the projection's epilogue, once per layer (head_dim 80 of JiT-H: in the attention kernel)
            if hdp != hd:  # a head size the kernels are not instantiated for: zero-padded heads (engine.ATTN_HEAD_DIMS)
                ones = torch.ones(hdp - hd)
                gains = tuple(bld.const(torch.cat([w_.detach().float().cpu(), ones])) for w_ in (at.q_norm.weight, at.k_norm.weight))
                wq, bq = engine.pad_qkv_heads(at.qkv.weight, at.qkv.bias, heads, hd, hdp, "3HC")
                qkv = bld.conv(n1, bld.pack_conv(wq, bq), 3 * heads * hdp)
                bld.free(n1)
                att = bld.attention(qkv, heads, "3HC", True, 1.0 / math.sqrt(hd), eps=1e-6, rope=rope_i, qk_weight=gains, norm_dim=hd)
                bld.free(qkv)
                x2 = bld.conv(att, bld.pack_conv(engine.pad_proj_heads(at.proj.weight, heads, hd, hdp), at.proj.bias), Hd,
                              gate=mod, gate_off=m0_ + 2 * Hd, gate_bstride=MS, res=x)
            else:
                qkv = bld.conv(n1, bld.pack_conv(at.qkv.weight, at.qkv.bias), 3 * Hd,
                               qk_prep=dict(heads=heads, head_dim=hd, rmsnorm=True, eps=1e-6, rope=rope_i, weight=gains))
                bld.free(n1)
                att = bld.attention(qkv, heads, "3HC", True, 1.0 / math.sqrt(hd), eps=1e-6, rope=rope_i, qk_weight=gains)
                bld.free(qkv)
                x2 = bld.conv(att, bld.pack_conv(at.proj.weight, at.proj.bias), Hd, gate=mod, gate_off=m0_ + 2 * Hd, gate_bstride=MS, res=x)
            bld.free(att)
            bld.free(x)
            n2 = bld.row_norm(x2, 1, weight=bld.const(blk.norm2.weight), scale=mod, shift=mod, scale_off=m0_ + 4 * Hd,
                              shift_off=m0_ + 3 * Hd, bstride=MS, eps=1e-6)
            # silu(x1) * x2 over halves (_src/model.py:159-162) -> interleave rows: pair (x2_c, x1_c)
            w12, b12 = blk.mlp.w12.weight.detach(), blk.mlp.w12.bias.detach()
            half = w12.shape[0] // 2
            w12i = torch.stack((w12[half:], w12[:half]), dim=1).reshape(2 * half, -1).contiguous()
            b12i = torch.stack((b12[half:], b12[:half]), dim=1).reshape(-1).contiguous()
            if half % 4 == 0:  # SwiGLU in the GEMM's epilogue (AzConvArgs.act = 4): no separate pass over the 4096-wide tensor
                glu = bld.conv(n2, bld.pack_conv(w12i, b12i), 2 * half, act=4)
                bld.free(n2)
            else:
                f1 = bld.conv(n2, bld.pack_conv(w12i, b12i), 2 * half)
                bld.free(n2)
                glu = bld.new_act(f1.B, f1.H, f1.W, half)
                tape.add("az_swiglu_f32", glu.ptr, f1.ptr, f1.B * f1.H * f1.W, half, f1.cs, glu.cs)
                bld.free(f1)
            x = bld.conv(glu, bld.pack_conv(blk.mlp.w3.weight, blk.mlp.w3.bias), Hd, gate=mod, gate_off=m0_ + 5 * Hd,
                         gate_bstride=MS, res=x2)
            bld.free(glu)
            bld.free(x2)
        if Lc and net.in_context_start < len(net.blocks):  # x[:, in_context_len:]
            body = bld.new_act(B, L, 1, Hd)
            tape.add("az_token_copy_f32", body.ptr, L, 0, x.ptr, L + Lc, Lc, L, B, Hd // 2 if body.half else Hd)
            bld.free(x)
            x = body

        # ---- final layer + unpatchify 'nhwpqc->nchpwq'          (_src/model.py:166-184,331-344)
        fl = net.final_layer
        mf = 6 * Hd * len(net.blocks)  # final layer's columns: shift | scale
        n = bld.row_norm(x, 1, weight=bld.const(fl.norm_final.weight), scale=mod, shift=mod, scale_off=mf + Hd, shift_off=mf,
                         bstride=MS, eps=1e-6)
        bld.free(x)
        wl = fl.linear.weight.detach().reshape(p * p, Z, Hd).transpose(0, 1).reshape(Z * p * p, Hd).contiguous()
        bl = fl.linear.bias.detach().reshape(p * p, Z).t().reshape(-1).contiguous()
        o = bld.conv(n, bld.pack_conv(wl, bl), Z * p * p, out_f32=True)  # (the plan's output: fp32 tokens for the unpatchify pass)
        bld.free(n)
        tape.add("az_unpatchify_f32", self.out.data_ptr(), o.ptr, B, Z, S, S, p, o.cs)
        bld.finish()
        self.tape = tape


class _Shared:
    r"""Marks a residual as batch-shared (``Builder.conv`` epilogue ``res_bcast``)."""

    def __init__(self, act: Act) -> None:
        self.act = act


class _TimestepEmbedder(nn.Module):
    def __init__(self, hidden: int, frequency_embedding_size: int = 256) -> None:
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden), nn.SiLU(), nn.Linear(hidden, hidden))
        self.frequency_embedding_size = frequency_embedding_size
        nn.init.normal_(self.mlp[0].weight, std=0.02)
        nn.init.normal_(self.mlp[2].weight, std=0.02)


class _LabelEmbedder(nn.Module):
    def __init__(self, num_classes: int, hidden: int) -> None:
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + 1, hidden)  # + the "null" class of CFG
        nn.init.normal_(self.embedding_table.weight, std=0.02)


class _PatchEmbed(nn.Module):
    def __init__(self, patch: int, channels: int, bottleneck: int, hidden: int) -> None:
        super().__init__()
        self.proj1 = nn.Conv2d(channels, bottleneck, kernel_size=patch, stride=patch, bias=False)
        self.proj2 = nn.Conv2d(bottleneck, hidden, kernel_size=1)
        nn.init.xavier_uniform_(self.proj1.weight.data.view(bottleneck, -1))
        nn.init.xavier_uniform_(self.proj2.weight.data.view(hidden, -1))
        nn.init.zeros_(self.proj2.bias)


class JiT(nn.Module):
    r"""Just image Transformer (reference ``_src/model.py:218-381``): class- and time-conditional pixel-space
    DiT with a bottleneck patch embedding, 6-way adaLN-zero blocks, SwiGLU FFN, q/k RMSNorm, 2-D rotary
    attention and ``in_context_len`` class tokens joining the sequence at block ``in_context_start``.

    ``forward(x, t, y)``: x (B, C, S, S), t (B,) or (1,), y (B,) int64 -> (B, C, S, S)."""

    def __init__(
        self,
        input_size: int = 256,
        patch_size: int = 16,
        in_channels: int = 3,
        hidden_size: int = 1024,
        depth: int = 24,
        num_heads: int = 16,
        mlp_ratio: float = 4.0,
        attn_drop: float = 0.0,
        proj_drop: float = 0.0,
        num_classes: int = 1000,
        bottleneck_dim: int = 128,
        in_context_len: int = 32,
        in_context_start: int = 8,
    ) -> None:
        super().__init__()
        if hidden_size % num_heads or hidden_size // num_heads > 128 or (hidden_size // num_heads) % 4:
            raise NotImplementedError(  # (the rotary layout splits a head into quarters: reference _src/util.py:100-143)
                f"head_dim {hidden_size / num_heads:g}: the gfx950 attention kernels take heads of up to 128 channels "
                "(16/32/64/80/128 natively, other multiples of 4 zero-padded to the next of those)"
            )
        if input_size % patch_size or hidden_size % 8:
            raise ValueError("input_size must be a multiple of patch_size and hidden_size of 8")
        self.in_channels = self.out_channels = in_channels
        self.patch_size, self.num_heads, self.hidden_size, self.input_size = patch_size, num_heads, hidden_size, input_size
        self.in_context_len, self.in_context_start, self.num_classes = in_context_len, in_context_start, num_classes
        self.t_embedder = _TimestepEmbedder(hidden_size)
        self.y_embedder = _LabelEmbedder(num_classes, hidden_size)
        self.x_embedder = _PatchEmbed(patch_size, in_channels, bottleneck_dim, hidden_size)
        grid = input_size // patch_size
        self.pos_embed = nn.Parameter(sincos_table(hidden_size, grid)[None], requires_grad=False)
        if in_context_len > 0:
            self.in_context_posemb = nn.Parameter(0.02 * torch.randn(1, in_context_len, hidden_size))
        self.blocks = nn.ModuleList([_Block(hidden_size, num_heads, mlp_ratio) for _ in range(depth)])
        self.final_layer = _Final(hidden_size, patch_size, in_channels)
        for blk in self.blocks:  # Xavier weights / zero biases; adaLN and the output layer start at zero
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.w12, blk.mlp.w3):
                nn.init.xavier_uniform_(lin.weight)
                nn.init.zeros_(lin.bias)
        self._plans: dict = {}

    def _param_versions(self) -> tuple:
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def plan(self, B: int, t_shared: bool, device) -> JiTPlan:
        key = (B, t_shared, str(device))
        p = self._plans.get(key)
        if p is None or p.versions != self._param_versions():
            p = self._plans[key] = JiTPlan(self, B, t_shared, device)
        return p

    def _check(self, x: Tensor) -> torch.dtype:
        from ...nn.utils import backbone_io_dtype

        out_dtype = backbone_io_dtype(self, x, "azula_amd.plugins.jit.JiT")
        if tuple(x.shape[1:]) != (self.in_channels, self.input_size, self.input_size):
            raise ValueError(f"expected (B, {self.in_channels}, {self.input_size}, {self.input_size}), got {tuple(x.shape)}")
        return out_dtype

    @torch.no_grad()
    @_lib.on_device
    def forward(self, x: Tensor, t: Tensor, y: Tensor) -> Tensor:
        out_dtype = self._check(x)
        B = x.shape[0]
        t = t.reshape(-1)
        plan = self.plan(B, t.numel() == 1, x.device)
        plan.x_nchw.copy_(x)
        plan.t.copy_(t.to(device=x.device, dtype=torch.float32))
        plan.labels.copy_(y.to(device=x.device, dtype=torch.int64).expand(B))
        plan.tape.run()
        return plan.out.to(out_dtype, copy=True)


_ARCH = {  # name: (depth, hidden, heads, bottleneck, context start, patch)   (_src/model.py:384-457)
    "JiT-B/16": (12, 768, 12, 128, 4, 16), "JiT-B/32": (12, 768, 12, 128, 4, 32),
    "JiT-L/16": (24, 1024, 16, 128, 8, 16), "JiT-L/32": (24, 1024, 16, 128, 8, 32),
    "JiT-H/16": (32, 1280, 16, 256, 10, 16), "JiT-H/32": (32, 1280, 16, 256, 10, 32),
}


def _factory(name: str):
    depth, hidden, heads, bottleneck, start, patch = _ARCH[name]

    def make(**kwargs) -> JiT:
        return JiT(depth=depth, hidden_size=hidden, num_heads=heads, bottleneck_dim=bottleneck, in_context_len=32,
                   in_context_start=start, patch_size=patch, **kwargs)

    return make


JiT_models = {name: _factory(name) for name in _ARCH}
