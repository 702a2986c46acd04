r"""Oracle: backbone forwards as pure functions over azula ``state_dict`` s (torch CPU, fp32).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Each network is restated as a function ``f(sd, cfg, x, ...)`` that walks the reference's
``state_dict`` keys and calls the same ATen ops the reference
modules dispatch, in the same order.  ``tap`` (optional dict) collects per-block outputs for
the per-layer fixtures (G5).
"""

from __future__ import annotations

import math
import torch
import torch.nn.functional as F

from torch import Tensor


# --------------------------------------------------------------------------- shared layers
def _linear(sd, key: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def _conv(sd, key: str, x: Tensor, stride=1, padding=None, periodic: bool = False) -> Tensor:
    r"""ConvNd (azula/nn/layers.py:25-50) for 1-D signals, 2-D images and 3-D volumes (B, C, *spatial); 'same' padding
    (kernel // 2 per axis, azula/nn/unet.py:170-172) unless given."""
    n = x.ndim - 2
    if padding is None:
        padding = tuple(k // 2 for k in sd[key + ".weight"].shape[2:])
        if len(set(padding)) == 1:
            padding = padding[0]
    if periodic and padding:
        assert isinstance(padding, int)  # torch.nn.ConvNd(padding_mode="circular"): F.pad(mode="circular") then an unpadded conv
        x = F.pad(x, (padding,) * (2 * n), mode="circular")
        padding = 0
    conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[n]
    return conv(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding)


def layer_norm_unbiased(x: Tensor, dim: int, eps: float = 1e-5) -> Tensor:
    r"""azula/nn/layers.py:152-155 (``var_mean`` default = unbiased, no affine)."""
    v, m = torch.var_mean(x, dim=dim, keepdim=True)
    return (x - m) * torch.rsqrt(v + eps)


def rms_norm(x: Tensor, dim: int = -1, eps: float = 1e-5) -> Tensor:
    r"""azula/nn/layers.py:193-195."""
    return x * torch.rsqrt(torch.mean(torch.square(x), dim=dim, keepdim=True) + eps)


def sine_encoding(x: Tensor, features: int, omega: float = 1e4) -> Tensor:
    r"""azula/nn/layers.py:286-299 (sin block then cos block)."""
    x = x.unsqueeze(dim=-1)
    freqs = torch.linspace(0, 1, features // 2, dtype=x.dtype)
    freqs = torch.exp(math.log(1 / omega) * freqs)
    return torch.cat((torch.sin(x * freqs), torch.cos(x * freqs)), dim=-1)


def _ada_zero(sd, key: str, mod: Tensor | None, channels: int, trailing: tuple[int, ...]):
    r"""(a, b, c) modulation triple -- azula/nn/unet.py:63-75,86-89, azula/nn/dit.py:57-68,102-105."""
    if key in sd:  # mod_features == 0: raw (3, C, 1, 1) / (3, C) parameter
        return sd[key].unbind(0)
    h = _linear(sd, key + ".0", mod)
    h = F.silu(h)
    h = _linear(sd, key + ".2", h)  # (..., 3C) laid out (n C)
    h = h.unflatten(-1, (3, channels)).movedim(-2, 0)  # n ... C
    return tuple(t.reshape(*t.shape[:-1], *trailing) for t in h.unbind(0))


# --------------------------------------------------------------------------- azula.nn.unet
def unet_block(sd, key: str, x: Tensor, mod, norm: str, groups: int, periodic: bool = False) -> Tensor:
    r"""UNetBlock._forward -- azula/nn/unet.py:85-95 (``periodic``: circular padding, unet.py:175-180)."""
    C = x.shape[1]
    a, b, c = _ada_zero(sd, key + ".ada_zero", mod, C, (C, *(1,) * (x.ndim - 2)))
    if norm == "group":
        n = F.group_norm(x, min(groups, C), eps=1e-5)  # affine=False, unet.py:54-60
    elif norm == "layer":
        n = layer_norm_unbiased(x, dim=1)  # LayerNorm(dim=-spatial-1): the channel axis
    elif norm == "rms":
        n = rms_norm(x, dim=1)
    else:
        raise NotImplementedError(norm)
    y = (a + 1) * n + b
    y = _conv(sd, key + ".ffn.0", y, periodic=periodic)
    y = F.silu(y)
    y = _conv(sd, key + ".ffn.3", y, periodic=periodic)
    return x + c * y


def unet_forward(sd, cfg: dict, x: Tensor, mod: Tensor | None = None, tap: dict | None = None):
    r"""UNet.forward -- azula/nn/unet.py:205-259 (wiring: :165-203).

    cfg keys: hid_channels, hid_blocks, norm ("group" | "layer" | "rms"), groups.
    """
    hid_blocks = list(cfg["hid_blocks"])
    norm, groups = cfg.get("norm", "layer"), cfg.get("groups", 16)
    per = cfg.get("periodic", False)
    stride = cfg.get("stride", 2)  # int or one per axis: the downsampling convolutions' stride = the nearest upsampling factor (:159-186)
    if not isinstance(stride, int):
        stride = tuple(stride)
    L = len(hid_blocks)
    memory = []
    for i in range(L):
        memory.append(x if memory else None)
        x = _conv(sd, f"descent.{i}.0", x, stride=stride if i > 0 else 1, periodic=per)
        for j in range(hid_blocks[i]):
            x = unet_block(sd, f"descent.{i}.{1 + j}", x, mod, norm, groups, per)
        if tap is not None:
            tap[f"descent.{i}"] = x
    for k in range(L):
        i = L - 1 - k  # ascent[0] is the deepest level
        idx = 0
        if i + 1 < L:
            x = _conv(sd, f"ascent.{k}.0", x, periodic=per)
            idx = 1
        for j in range(hid_blocks[i]):
            x = unet_block(sd, f"ascent.{k}.{idx + j}", x, mod, norm, groups, per)
        idx += hid_blocks[i]
        if i > 0:
            x = F.interpolate(x, scale_factor=tuple(float(v) for v in stride) if isinstance(stride, tuple) else (float(stride),) * (x.ndim - 2), mode="nearest")
        else:
            x = _conv(sd, f"ascent.{k}.{idx}", x, periodic=per)
        if tap is not None:
            tap[f"ascent.{k}"] = x
        y = memory.pop()
        if y is None:
            continue
        for d in range(2, x.ndim):
            if x.shape[d] > y.shape[d]:
                x = torch.narrow(x, d, 0, y.shape[d])
        x = torch.cat((y, x), dim=1)
    return x


def time_wrapped_unet(sd, cfg: dict, x: Tensor, c_time: Tensor, **_) -> Tensor:
    r"""The tutorial's time-embedding wrapper (docs/tutorials/mnist.ipynb cell 8, no label):
    ``mod = Linear(1, D) -> SiLU -> Linear(D, D)`` on ``c_time[..., None]`` then ``unet(x, mod)``.
    Keys: ``time_embedding.{0,2}``, ``unet.*``."""
    mod = _linear(sd, "time_embedding.0", c_time[..., None])
    mod = _linear(sd, "time_embedding.2", F.silu(mod))
    sub = {k[len("unet."):]: v for k, v in sd.items() if k.startswith("unet.")}
    return unet_forward(sub, cfg, x, mod)


# --------------------------------------------------------------------------- azula.nn.dit / vit
def apply_rope(q: Tensor, k: Tensor, theta: Tensor):
    r"""azula/nn/attention.py:112-156: rotate adjacent (real, imag) channel pairs by theta."""

    def rot(v):
        v = v.unflatten(-1, (-1, 2))
        re, im = v[..., 0], v[..., 1]
        c, s_ = torch.cos(theta), torch.sin(theta)
        return torch.stack((re * c - im * s_, re * s_ + im * c), dim=-1).flatten(-2)

    return rot(q), rot(k)


def msa_forward(sd, key: str, x: Tensor, heads: int, qk_norm: bool = True, pos: Tensor | None = None,
                mask: Tensor | None = None) -> Tensor:
    r"""MultiheadSelfAttention.forward -- azula/nn/attention.py:89-108 (RoPE iff ``theta_proj`` exists; ``mask`` is
    handed to scaled_dot_product_attention as ``attn_mask``)."""
    qkv = _linear(sd, key + ".qkv_proj", x)  # (..., L, (n H C))
    *lead, L, _ = qkv.shape
    qkv = qkv.reshape(*lead, L, 3, heads, -1)
    q, k, v = (qkv[..., n, :, :].transpose(-2, -3) for n in range(3))  # ... H L C
    if qk_norm:
        C = q.shape[-1]
        q = F.rms_norm(q, (C,), eps=1e-5)  # torch.nn.RMSNorm path, attention.py:48-56
        k = F.rms_norm(k, (C,), eps=1e-5)
    if key + ".theta_proj.weight" in sd:
        theta = F.linear(pos, sd[key + ".theta_proj.weight"])  # ... L (H C)
        theta = theta.unflatten(-1, (heads, -1)).transpose(-2, -3)  # ... H L C
        q, k = apply_rope(q, k, theta)
    y = F.scaled_dot_product_attention(query=q, key=k, value=v, attn_mask=mask)
    y = y.transpose(-2, -3).flatten(-2)  # ... L (H C)
    return F.linear(y, sd[key + ".y_proj.weight"])


def dit_block(sd, key: str, x: Tensor, mod, heads: int, pos=None, act: str = "silu", qk_norm: bool = True, mask=None) -> Tensor:
    r"""DiTBlock._forward -- azula/nn/dit.py:95-112 (FFN activations: dit.py:74-85, layers.py:71-110)."""
    C = x.shape[-1]
    a, b, c = _ada_zero(sd, key + ".ada_zero", mod, C, (1, C))
    y = (a + 1) * F.rms_norm(x, (C,), eps=1e-5) + b
    y = y + msa_forward(sd, key + ".msa", y, heads, qk_norm=qk_norm, pos=pos, mask=mask)
    y = _linear(sd, key + ".ffn.0", y)
    if act == "silu":
        y = F.silu(y)
    elif act == "relu":
        y = F.relu(y)
    elif act == "relu2":
        y = F.relu(y).square()
    elif act == "swiglu":
        y = y.unflatten(-1, (-1, 2))
        y = y[..., 0] * F.silu(y[..., 1])
    y = _linear(sd, key + ".ffn.3", y)
    return x + c * y


def dit_forward(sd, cfg: dict, x: Tensor, mod=None, pos: Tensor | None = None, tap=None):
    r"""DiT.forward -- azula/nn/dit.py:184-218.  cfg: hid_channels, hid_blocks, attention_heads."""
    x = _linear(sd, "in_proj", x)
    if pos is None:
        pos = torch.arange(x.shape[-2], dtype=x.dtype)[..., None]
    e = sine_encoding(pos, cfg["hid_channels"], omega=1e2).flatten(-2)  # ... (P C)
    x = x + F.linear(e, sd["pos_embedding.2.weight"])
    for i in range(cfg["hid_blocks"]):
        x = dit_block(sd, f"blocks.{i}", x, mod, cfg["attention_heads"], pos=pos, act=cfg.get("ffn_activation", "silu"),
                      qk_norm=cfg.get("qk_norm", True))
        if tap is not None:
            tap[f"blocks.{i}"] = x
    return _linear(sd, "out_proj", x)


def _patchify(x: Tensor, patch) -> Tensor:
    r"""'... Z (A a) (B b) ... -> ... A B ... (Z a b ...)' -- azula/nn/layers.py:198-222 (channel_last), any number of
    spatial dimensions."""
    B, Z, *dims = x.shape
    n = len(patch)
    x = x.reshape(B, Z, *[v for d, p in zip(dims, patch) for v in (d // p, p)])
    x = x.permute(0, *[2 + 2 * i for i in range(n)], 1, *[3 + 2 * i for i in range(n)])
    return x.reshape(B, *[d // p for d, p in zip(dims, patch)], Z * math.prod(patch))


def _unpatchify(y: Tensor, patch) -> Tensor:
    r"""The inverse rearrangement -- azula/nn/layers.py:225-247."""
    B, *grid, Fe = y.shape
    n = len(patch)
    Z = Fe // math.prod(patch)
    y = y.reshape(B, *grid, Z, *patch)
    return y.permute(0, 1 + n, *[v for i in range(n) for v in (1 + i, 2 + n + i)]).reshape(
        B, Z, *[g * p for g, p in zip(grid, patch)])


def vit_forward(sd, cfg: dict, x: Tensor, mod=None, tap=None, cond: Tensor | None = None) -> Tensor:
    r"""ViT.forward -- azula/nn/vit.py:76-108 (Patchify/Unpatchify channel_last, azula/nn/layers.py:198-247).  cfg adds
    patch_size / optionally unpatch_size (int or one entry per spatial dimension); ``cond`` is patchified like ``x`` and
    concatenated along the token features (vit.py:92-96, dit.py:202-203)."""
    n = x.ndim - 2
    as_shape = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v,) * n  # noqa: E731
    p = as_shape(cfg["patch_size"])
    pu = as_shape(cfg.get("unpatch_size") or cfg["patch_size"])
    t = _patchify(x, p)
    shape = t.shape[1:-1]
    pos = torch.cartesian_prod(*(torch.arange(s, dtype=x.dtype) for s in shape)).reshape(-1, len(shape))
    t = t.flatten(1, -2)
    if cond is not None:
        t = torch.cat((t, _patchify(cond, p).flatten(1, -2)), dim=-1)
    y = dit_forward(sd, cfg, t, mod, pos=pos, tap=tap)
    return _unpatchify(y.unflatten(-2, shape), pu)


def time_wrapped_vit(sd, cfg: dict, x: Tensor, c_time: Tensor, **_) -> Tensor:
    r"""Same wrapper pattern as :func:`time_wrapped_unet` around a ViT (keys ``vit.*``)."""
    mod = _linear(sd, "time_embedding.0", c_time[..., None])
    mod = _linear(sd, "time_embedding.2", F.silu(mod))
    sub = {k[len("vit."):]: v for k, v in sd.items() if k.startswith("vit.")}
    return vit_forward(sub, cfg, x, mod)


# --------------------------------------------------------------------------- ADM UNetModel
def adm_timestep_embedding(timesteps: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    r"""cos || sin, fp32 -- azula/plugins/adm/_src/nn.py:90-108."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn32(sd, key: str, x: Tensor) -> Tensor:
    return F.group_norm(x, 32, sd[key + ".weight"], sd[key + ".bias"], eps=1e-5)  # _src/nn.py:80-87


def adm_upsample(x: Tensor) -> Tensor:
    r"""Upsample.forward without its convolution -- _src/unet.py:101-106: nearest x2; volumes keep their depth."""
    if x.ndim == 5:
        return F.interpolate(x, (x.shape[2], x.shape[3] * 2, x.shape[4] * 2), mode="nearest")
    return F.interpolate(x, scale_factor=2, mode="nearest")


def adm_down_stride(x: Tensor):
    r"""``stride = 2 if dims != 3 else (1, 2, 2)`` -- _src/unet.py:128."""
    return (1, 2, 2) if x.ndim == 5 else 2


def adm_avg_pool(x: Tensor) -> Tensor:
    r"""avg_pool_nd(dims, kernel_size=stride, stride=stride) -- _src/nn.py:64-77, _src/unet.py:133."""
    st = adm_down_stride(x)
    return {3: F.avg_pool1d, 4: F.avg_pool2d, 5: F.avg_pool3d}[x.ndim](x, st, st)


def adm_resblock(sd, key: str, x: Tensor, emb: Tensor, up: bool, down: bool, scale_shift: bool) -> Tensor:
    r"""ResBlock._forward -- azula/plugins/adm/_src/unet.py:227-247 (dims = 1: (B, C, L) signals, dims = 2: images, dims = 3:
    volumes, whose depth axis is never resampled: ``Upsample`` / ``Downsample`` act on the inner two axes, :103-104,128)."""
    n = x.ndim - 2
    h = F.silu(_gn32(sd, key + ".in_layers.0", x))
    if up:  # :104-106 nearest x2 on both branches
        h, x = adm_upsample(h), adm_upsample(x)
    elif down:  # :133 AvgPoolNd(2, 2)
        h, x = adm_avg_pool(h), adm_avg_pool(x)
    h = _conv(sd, key + ".in_layers.2", h)
    emb_out = _linear(sd, key + ".emb_layers.1", F.silu(emb))[(...,) + (None,) * n]  # :236-237
    if scale_shift:
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = _gn32(sd, key + ".out_layers.0", h) * (1 + scale) + shift
        h = _conv(sd, key + ".out_layers.3", F.silu(h))
    else:
        h = h + emb_out
        h = _conv(sd, key + ".out_layers.3", F.silu(_gn32(sd, key + ".out_layers.0", h)))
    if key + ".skip_connection.weight" in sd:
        x = _conv(sd, key + ".skip_connection", x)
    return x + h


def adm_attention(sd, key: str, x: Tensor, heads: int, new_order: bool) -> Tensor:
    r"""AttentionBlock._forward + QKVAttention(Legacy) -- _src/unet.py:289-296,330-384."""
    b, c, *spatial = x.shape
    x = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn32(sd, key + ".norm", x), sd[key + ".qkv.weight"], sd[key + ".qkv.bias"])
    bs, width, length = qkv.shape
    ch = width // (3 * heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = (q * scale).view(bs * heads, ch, length)
        k = (k * scale).view(bs * heads, ch, length)
        v = v.reshape(bs * heads, ch, length)
        w = torch.einsum("bct,bcs->bts", q, k)
    else:
        q, k, v = qkv.reshape(bs * heads, ch * 3, length).split(ch, dim=1)
        w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)
    h = F.conv1d(a, sd[key + ".proj_out.weight"], sd[key + ".proj_out.bias"])
    return (x + h).reshape(b, c, *spatial)


def adm_layout(cfg: dict):
    r"""Block list of UNetModel.__init__ -- _src/unet.py:468-600.  Returns
    (input_blocks, middle, output_blocks); each block is a list of
    ("res", up, down) | ("attn", heads) | ("conv",) | ("down", use_conv) | ("up", use_conv) layer descriptors."""
    mc, mult = cfg["num_channels"], cfg["channel_mult"]
    nres = cfg["num_res_blocks"]
    attn_ds = {cfg["image_size"] // r for r in cfg["attention_resolutions"]}  # plugins/adm/__init__.py:182
    updown = cfg.get("resblock_updown", False)
    use_conv = cfg.get("conv_resample", True)  # Down/Upsample layers of resblock_updown=False, _src/unet.py:516-518,591-593
    hc = cfg.get("num_head_channels", -1)

    def heads(ch):
        return cfg.get("num_heads", 1) if hc == -1 else ch // hc

    ch = int(mult[0] * mc)
    inputs, chans, ds = [[("conv",)]], [ch], 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            ch = int(m * mc)
            blk = [("res", False, False)]
            if ds in attn_ds:
                blk.append(("attn", heads(ch)))
            inputs.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inputs.append([("res", False, True)] if updown else [("down", use_conv)])
            chans.append(ch)
            ds *= 2
    middle = [("res", False, False), ("attn", heads(ch)), ("res", False, False)]
    outputs = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            chans.pop()
            ch = int(mc * m)
            blk = [("res", False, False)]
            if ds in attn_ds:
                blk.append(("attn", heads(ch)))
            if level and i == nres:
                blk.append(("res", True, False) if updown else ("up", use_conv))
                ds //= 2
            outputs.append(blk)
    return inputs, middle, outputs


def adm_unet_forward(sd, cfg: dict, x: Tensor, timesteps: Tensor, y: Tensor | None = None, tap=None):
    r"""UNetModel.forward -- azula/plugins/adm/_src/unet.py:605-634."""
    inputs, middle, outputs = adm_layout(cfg)
    ss = cfg.get("use_scale_shift_norm", False)
    new_order = cfg.get("use_new_attention_order", False)
    emb = adm_timestep_embedding(timesteps, cfg["num_channels"])
    emb = _linear(sd, "time_embed.2", F.silu(_linear(sd, "time_embed.0", emb)))
    if cfg.get("num_classes") is not None:
        assert y is not None and y.shape == (x.shape[0],)
        emb = emb + F.embedding(y, sd["label_emb.weight"])
    else:
        assert y is None

    def run(prefix, blk, h):
        for j, layer in enumerate(blk):
            key = f"{prefix}.{j}"
            if layer[0] == "conv":
                h = _conv(sd, key, h)
            elif layer[0] == "res":
                h = adm_resblock(sd, key, h, emb, layer[1], layer[2], ss)
            elif layer[0] == "down":  # Downsample.forward, _src/unet.py:128-137: stride-2 3-tap conv or AvgPoolNd(2, 2)
                h = _conv(sd, key + ".op", h, stride=adm_down_stride(h)) if layer[1] else adm_avg_pool(h)
            elif layer[0] == "up":  # Upsample.forward, _src/unet.py:101-109: nearest x2, then the optional 3x3 conv
                h = adm_upsample(h)
                if layer[1]:
                    h = _conv(sd, key + ".conv", h)
            else:
                h = adm_attention(sd, key, h, layer[1], new_order)
        return h

    hs, h = [], x
    for i, blk in enumerate(inputs):
        h = run(f"input_blocks.{i}", blk, h)
        hs.append(h)
        if tap is not None:
            tap[f"input_blocks.{i}"] = h
    h = run("middle_block", middle, h)
    if tap is not None:
        tap["middle_block"] = h
    for i, blk in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(f"output_blocks.{i}", blk, h)
        if tap is not None:
            tap[f"output_blocks.{i}"] = h
    h = F.silu(_gn32(sd, "out.0", h))
    return _conv(sd, "out.2", h)


# --------------------------------------------------------------------------- synthetic weights
def rerandomise_zero_tensors(sd: dict, seed: int = 123) -> int:
    r"""Refill every all-zero weight tensor with ndim > 1 by N(0, 1/fan_in) from
    ``torch.Generator().manual_seed(seed)``: random-init ADM has 84 such
    tensors (``zero_module``, _src/unet.py:207,285,602) which would make parity vacuous."""
    g = torch.Generator().manual_seed(seed)
    n = 0
    for k in sorted(sd):
        v = sd[k]
        if torch.is_floating_point(v) and v.ndim > 1 and not torch.any(v != 0):
            fan_in = v[0].numel()
            v.copy_(torch.randn(v.shape, generator=g) / math.sqrt(fan_in))
            n += 1
    return n


# --------------------------------------------------------------------------- JiT plugin backbone
JIT_ARCH = {  # plugins/jit/_src/model.py:392-467 (depth, hidden, heads, bottleneck, ctx len, ctx start, patch)
    "JiT-B/16": (12, 768, 12, 128, 32, 4, 16), "JiT-B/32": (12, 768, 12, 128, 32, 4, 32),
    "JiT-L/16": (24, 1024, 16, 128, 32, 8, 16), "JiT-L/32": (24, 1024, 16, 128, 32, 8, 32),
    "JiT-H/16": (32, 1280, 16, 256, 32, 10, 16), "JiT-H/32": (32, 1280, 16, 256, 32, 10, 32),
}


def jit_rms_norm(x: Tensor, weight: Tensor, eps: float = 1e-6) -> Tensor:
    r"""weight * x * rsqrt(mean(x^2) + eps), statistics in fp32 -- plugins/jit/_src/util.py:149-163."""
    h = x.to(torch.float32)
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return (weight * h).to(x.dtype)


def jit_timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    r"""cos block || sin block -- plugins/jit/_src/model.py:59-81."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(t.dtype)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def jit_rope_tables(head_dim: int, grid: int, ctx: int = 0, theta: float = 10000.0):
    r"""(cos, sin), each (ctx + grid^2, head_dim): the first half of a head rotates with the row index,
    the second half with the column index, adjacent channel pairs share an angle; context tokens
    are not rotated (cos 1, sin 0) -- plugins/jit/_src/util.py:100-143 with dim = head_dim / 2."""
    dim = head_dim // 2
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    pos = torch.arange(grid) / grid * grid
    ang = (pos[:, None] * freqs[None]).repeat_interleave(2, dim=-1)  # (grid, dim)
    full = torch.cat((ang[:, None, :].expand(grid, grid, dim), ang[None, :, :].expand(grid, grid, dim)), dim=-1)
    full = full.reshape(grid * grid, head_dim)
    cos, sin = full.cos(), full.sin()
    if ctx > 0:
        cos = torch.cat([torch.ones(ctx, head_dim), cos], dim=0)
        sin = torch.cat([torch.zeros(ctx, head_dim), sin], dim=0)
    return cos, sin


def jit_rotate(t: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    r"""t cos + rotate_half(t) sin over adjacent pairs (x0, x1) -> (-x1, x0) -- util.py:32-36,145-146."""
    pairs = t.unflatten(-1, (-1, 2))
    rot = torch.stack((-pairs[..., 1], pairs[..., 0]), dim=-1).flatten(-2)
    return t * cos + rot * sin


def jit_pos_embed(hidden: int, grid: int) -> Tensor:
    r"""Fixed 2-D sin-cos table (grid^2, hidden), fp64 -> fp32 -- plugins/jit/_src/util.py:166-212
    (column coordinate feeds the first half: "here w goes first")."""
    import numpy as np

    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    gw, gh = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32))
    emb = np.concatenate([one_d(hidden // 2, gw), one_d(hidden // 2, gh)], axis=1)
    return torch.from_numpy(emb).float()


def jit_attention(sd, key: str, x: Tensor, heads: int, cos: Tensor, sin: Tensor) -> Tensor:
    r"""Attention.forward -- plugins/jit/_src/model.py:121-147."""
    B, N, C = x.shape
    qkv = _linear(sd, key + ".qkv", x).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = jit_rotate(jit_rms_norm(q, sd[key + ".q_norm.weight"]), cos, sin)
    k = jit_rotate(jit_rms_norm(k, sd[key + ".k_norm.weight"]), cos, sin)
    y = F.scaled_dot_product_attention(q, k, v)
    return _linear(sd, key + ".proj", y.transpose(1, 2).reshape(B, N, C))


def jit_block(sd, key: str, x: Tensor, c: Tensor, heads: int, cos: Tensor, sin: Tensor) -> Tensor:
    r"""JiTBlock.forward -- plugins/jit/_src/model.py:205-215; SwiGLUFFN :150-163."""
    mod = _linear(sd, key + ".adaLN_modulation.1", F.silu(c)).chunk(6, dim=-1)
    shift_a, scale_a, gate_a, shift_m, scale_m, gate_m = (m.unsqueeze(1) for m in mod)
    h = jit_rms_norm(x, sd[key + ".norm1.weight"]) * (1 + scale_a) + shift_a
    x = x + gate_a * jit_attention(sd, key + ".attn", h, heads, cos, sin)
    h = jit_rms_norm(x, sd[key + ".norm2.weight"]) * (1 + scale_m) + shift_m
    x1, x2 = _linear(sd, key + ".mlp.w12", h).chunk(2, dim=-1)
    return x + gate_m * _linear(sd, key + ".mlp.w3", F.silu(x1) * x2)


def jit_forward(sd, cfg: dict, x: Tensor, t: Tensor, y: Tensor, tap: dict | None = None) -> Tensor:
    r"""JiT.forward -- plugins/jit/_src/model.py:346-381.  ``cfg``: {model, input_size} or explicit
    {depth, hidden_size, num_heads, bottleneck_dim, in_context_len, in_context_start, patch_size, input_size}."""
    if "model" in cfg:
        depth, hidden, heads, _, ctx, start, p = JIT_ARCH[cfg["model"]]
    else:
        depth, hidden, heads, ctx, start, p = (cfg[k] for k in (
            "depth", "hidden_size", "num_heads", "in_context_len", "in_context_start", "patch_size"))
    grid = cfg.get("input_size", 256) // p
    hd = hidden // heads
    t_emb = jit_timestep_embedding(t, 256)
    t_emb = _linear(sd, "t_embedder.mlp.2", F.silu(_linear(sd, "t_embedder.mlp.0", t_emb)))
    y_emb = sd["y_embedder.embedding_table.weight"][y]
    c = t_emb + y_emb
    h = F.conv2d(x, sd["x_embedder.proj1.weight"], None, stride=p)
    h = F.conv2d(h, sd["x_embedder.proj2.weight"], sd["x_embedder.proj2.bias"]).flatten(2).transpose(1, 2)
    h = h + sd["pos_embed"]
    rope_img = jit_rope_tables(hd, grid, 0)
    rope_ctx = jit_rope_tables(hd, grid, ctx)
    for i in range(depth):
        if ctx > 0 and i == start:
            tokens = y_emb.unsqueeze(1).repeat(1, ctx, 1) + sd["in_context_posemb"]
            h = torch.cat([tokens, h], dim=1)
        h = jit_block(sd, f"blocks.{i}", h, c, heads, *(rope_img if i < start else rope_ctx))
        if tap is not None:
            tap[f"block{i}"] = h
    h = h[:, ctx:]
    shift, scale = _linear(sd, "final_layer.adaLN_modulation.1", F.silu(c)).chunk(2, dim=1)
    h = jit_rms_norm(h, sd["final_layer.norm_final.weight"]) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    h = _linear(sd, "final_layer.linear", h)  # (B, grid^2, p p C) with feature order (p, q, c)
    B, C = x.shape[0], x.shape[1]
    h = h.reshape(B, grid, grid, p, p, C)
    return torch.einsum("nhwpqc->nchpwq", h).reshape(B, C, grid * p, grid * p)
