r"""Host-side execution engine: kernel tapes, static activation pools and hipGraph replay.

A backbone forward is compiled ONCE per input shape into a :class:`Tape` -- a flat list of
C-ABI calls whose every argument (device pointers, sizes, POD structs) is static.  Per-step
scalars live in device memory (``AzStepCoef``), so the same tape -- and the hipGraph captured
from it -- serves every sampling step.  Python is only the builder; replay is one
``hipGraphLaunch`` per step.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import AzConvArgs, AzNormFinalizeArgs, AzTransitionArgs


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


# Modules cast to half precision (.bfloat16() / .half()): "1" (default since round 6) = their activations live in HBM in the module's
# own type between the layers, as in the reference (azula/denoise.py:314-320 casts the backbone input to the module's dtype, so every
# tensor of the forward is a half tensor); statistics, softmax, gates and residual ADDS are evaluated in fp32 registers.  "0" = fp32
# activations in HBM, converted per tile (rounds 2 - 5).  Only plans whose every kernel has the typed form take it (ViT / DiT).
HALF_ACT = os.environ.get("AZ_HALF_ACT", "1") != "0"


class Tape:
    r"""A recorded sequence of C-ABI kernel launches with static arguments."""

    def __init__(self) -> None:
        self.ops: list[tuple] = []
        self.keep: list = []  # tensors / structs that must outlive the tape

    def add(self, name: str, *args, keep=()) -> None:
        fn = getattr(_lib.lib(), name)
        self.ops.append((fn, args, name))
        self.keep.extend(keep)

    def extend(self, other: "Tape") -> None:
        self.ops.extend(other.ops)
        self.keep.extend(other.keep)

    def run(self, stream: int | None = None) -> None:
        if stream is None:
            stream = _lib.stream_ptr()
        for fn, args, name in self.ops:
            rc = fn(*args, stream)
            if rc != 0:
                _lib.check(rc, name)

    def __len__(self) -> int:
        return len(self.ops)


class StepGraph:
    r"""hipGraph captured from a tape (on a private capture stream), launched on torch's stream."""

    def __init__(self, tape: Tape, device: torch.device) -> None:
        self.tape = tape
        self.handle = C.c_void_p()
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            _lib.call("az_graph_begin", side.cuda_stream)
            try:
                tape.run(side.cuda_stream)
            finally:
                _lib.call("az_graph_end", side.cuda_stream, C.byref(self.handle))
        torch.cuda.current_stream(device).wait_stream(side)
        self._side = side

    def launch(self) -> None:
        _lib.call("az_graph_launch", self.handle, _lib.stream_ptr())

    @property
    def num_nodes(self) -> int:
        n = C.c_int64()
        _lib.call("az_graph_num_nodes", self.handle, C.byref(n))
        return n.value

    def __del__(self) -> None:
        try:
            if self.handle:
                _lib.lib().az_graph_destroy(self.handle)
        except Exception:
            pass


class Pool:
    r"""Static activation buffers with explicit reuse (a graph needs fixed addresses)."""

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.free: dict[int, list[torch.Tensor]] = {}
        self.all: list[torch.Tensor] = []

    def alloc(self, numel: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        lst = self.free.get((numel, dtype))
        if lst:
            return lst.pop()
        t = torch.empty(numel, dtype=dtype, device=self.device)
        self.all.append(t)
        return t

    def release(self, t: torch.Tensor) -> None:
        self.free.setdefault((t.numel(), t.dtype), []).append(t)

    @property
    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.all)


class Act:
    r"""An NHWC activation: ``buf`` holds (B, H, W, cs) floats, ``C`` real channels."""

    __slots__ = ("buf", "B", "H", "W", "C", "cs", "pinned", "gn_quads", "affine", "qk_prepared", "bounded", "absmax")

    @property
    def half(self) -> bool:
        r"""The buffer holds a 2-byte type (the activations of a module cast to half precision, engine.HALF_ACT)."""
        return self.buf.dtype != torch.float32

    def __init__(self, buf: torch.Tensor, B: int, H: int, W: int, C_: int, cs: int, pinned: bool = False) -> None:
        self.buf, self.B, self.H, self.W, self.C, self.cs, self.pinned = buf, B, H, W, C_, cs, pinned
        # bounded: the magnitudes in this tensor do not scale with the sampler's state -- it is the output of a normalisation, or of
        # convolutions / attention over such outputs without a residual add (bounded by the WEIGHTS).  The f16x2 kernels, whose
        # activation operand has a stated range (|x| < ~1e6, include/azula_amd.h), are chosen only for bounded inputs; residual /
        # input streams (which grow with x_t: an unstable multistep sampler reaches 1e7 in tests/test_gpu_unet.py) go through
        # the bf16x3 kernels, whose domain is all of fp32.  Default False: unknown = unbounded.
        self.bounded = False
        self.absmax = None  # AZ_ABSMAX_SLOTS floats written by az_absmax_f32 over this tensor on the tape so far (Builder.absmax_of), or None
        self.gn_quads = None  # (partials tensor, chunks per image): GroupNorm moments written by the producing conv
        self.qk_prepared = False  # a fused qkv projection whose q / k are already normalised / gained / rotated (AzConvArgs.act = 5)
        self.affine = None    # ([scale | shift] tensor, act): a normalisation whose apply pass has not run -- the values are
        #                       act(buf * scale + shift); Builder.conv evaluates it inside the Winograd gather or materialises it

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()


# Winograd for stride-1 3x3 convs (see conv.hip): "1" = exact F(2x2,3x3) where it pays (default), "0" = never,
# "2" = F(2x2) wherever it is legal (used by the parity tests to push whole networks through it),
# "4" = the faster but inexact F(4x4,3x3) on layers with >= WINOGRAD4_MIN_TILES tiles, "1" elsewhere
WINOGRAD = os.environ.get("AZ_WINOGRAD", "1")
# How fp32 convolutions / token GEMMs that run on the DIRECT kernel (1 x 1 convolutions, token linears, stride-2 and small-map
# 3 x 3) use the matrix pipe.  "bf16x3" (default since round 4, by the round-3 reviewer's ruling): every fp32 operand split
# EXACTLY into three bf16 pieces, the six largest partial products accumulated in fp32 on v_mfma_f32_32x32x16_bf16
# (az_conv2d_x3_f32) -- measured MORE accurate against fp64 than the fp32 MFMA (tests/test_gpu_kernels.py::
# test_conv2d_x3_accuracy) at 0.375 x its matrix-pipe time: DiT-B/2 54.7 -> 70.1 images/s, JiT-B/16 43.6 -> 58.4.
# "native": v_mfma_f32_32x32x2_f32 everywhere.  The stride-1 3 x 3 convolutions run the Winograd kernel in both modes (bf16x3: WINO_X3 below).
# "f16x2" (default since round 6): every fp32 operand as TWO IEEE half pieces (activations: h and the residual scaled by 2^11, of
# x / 16; weights: wh, wl and wh / 2^11 of w times a power of two fixed at pack time), three partial products on
# v_mfma_f32_32x32x16_f16 in one fp32 accumulator -- half the matrix instructions of bf16x3, measured the MOST accurate of the
# modes against fp64 (half as many fp32 accumulation steps: rms 5.2e-7 against 6.7e-7 bf16x3 / 7.5e-7 fp32 MFMA on the K = 2304
# layer of test_conv2d_x3_accuracy), on a STATED range of its activation operand, |x| < ~1e6 (beyond it NaN, never a wrong finite
# value: include/azula_amd.h).  It therefore takes only BOUNDED inputs (Act.bounded: outputs of normalisations and of
# convolutions / attention over them); layers that read the residual / input stream run the bf16x3 kernels in this mode too.
# C2 16.95 -> 15.2 ms per denoise step, C3 85 -> 107+ images/s, C5 32.1 -> 28.5 ms, C6 63.6 -> 84 images/s (profiles/r06_f16x2_gate*.txt).
FP32_MFMA = os.environ.get("AZ_FP32_MFMA", "f16x2")
assert FP32_MFMA in ("native", "bf16x3", "f16x2"), FP32_MFMA


def pieces() -> bool:
    r"""fp32 operands as 2-byte pieces on the bf16 / f16 pipe (read per plan: bench.py flips FP32_MFMA between plans)."""
    return FP32_MFMA in ("bf16x3", "f16x2")


# f16x2 mode, layers whose input is NOT bounded (residual / input streams): "1" (default) = f16x2 kernels too where it pays, with the
# activation scale taken from the sources' largest magnitude (one streaming az_absmax_f32 pass per source tensor and step, shared
# by its consumers; AzConvArgs.in_absmax0 / in_absmax1: no stated range) -- the UNet's strided and skip-merge convolutions; "0" = bf16x3
F16X2_DYNAMIC = os.environ.get("AZ_F16X2_DYNAMIC", "1") != "0"
# ... and where the tensor's producer left GroupNorm moments (Act.gn_quads), the maximum is BOUNDED from them instead of measured
# (az_absmax_from_moments_f32: no pass over the tensor) -- ADM's 1x1 skip projections, whose pass would cost what it saves ("0": measure)
F16X2_MOMENTS = os.environ.get("AZ_F16X2_MOMENTS", "1") != "0"
ATTN_H2 = os.environ.get("AZ_ATTN_H2", "1") != "0"  # f16x2 mode: the attention contractions in that form too ("0": bf16x3 attention -- A/B)
ATTN_X3 = os.environ.get("AZ_ATTN_X3", "1") != "0"  # bf16x3 mode: attention contractions on the bf16 pipe too (az_attention_x3_f32)
# The stride-1 3 x 3 layers in bf16x3 mode: "1" (default since round 5) = the Winograd kernel with its 16 frequency GEMMs on the bf16 pipe
# as exact 3 x bf16 splits too (az_conv2d_winograd_x3_f32, csrc/wino_x3.hip: same transforms, same epilogue, 1.19 - 1.28 x the fp32
# stream on the UNet / ADM layers, profiles/r05_wx3_*); "0" = the fp32-MFMA Winograd stream (always the one in "native" mode).
WINO_X3 = os.environ.get("AZ_WINO_X3", "1") != "0"
X3_MIN_CHANNELS = 32  # bf16x3 only where both channel counts fill a K tile / an MFMA tile
WINOGRAD4_MIN_TILES = int(os.environ.get("AZ_WINOGRAD4_MIN_TILES", "1024"))
# q / k RMS norm, gains and RoPE of an attention layer in the epilogue of its qkv projection ("0": inside the attention kernel)
QK_PREP = os.environ.get("AZ_QK_PREP", "1") != "0"
# GroupNorm statistics from the producing convolution's epilogue ("1", default) or always by the separate pass ("0")
GN_FUSED = os.environ.get("AZ_GN_FUSED", "1") != "0"
# GroupNorm apply pass (y = x * S + T) inside the consuming Winograd convolution's gather ("0": always the separate pass)
# The first convolution reads the latent planar (az_conv2d_stem_f32) instead of an NHWC copy of it ("0": the NHWC path)
STEM_PLANAR = os.environ.get("AZ_STEM_PLANAR", "1") != "0"
AFFINE_FUSED = os.environ.get("AZ_AFFINE_FUSED", "1") != "0"
GN_FUSED_SPLITK = os.environ.get("AZ_GN_FUSED", "1") != "epilogue"  # ("epilogue": only the Winograd epilogue's moments -- A/B)


WINO_X3_NAMES = ("az_conv2d_winograd_x3_f32", "az_conv2d_winograd_f16x2_f32")  # wino_x3.hip: the piece forms of the Winograd kernel


class ConvWeights:
    r"""Convolution / linear weights in the layouts the kernels consume."""

    def __init__(self, bld: "Builder", weight: torch.Tensor, bias: torch.Tensor | None, cin0: int | None) -> None:
        w = weight.detach().to(device=bld.device, dtype=torch.float32)
        if w.ndim == 2:
            w = w[:, :, None, None]
        elif w.ndim == 3 and w.shape[2] == 1:  # Conv1d k=1 (ADM attention projections)
            w = w[:, :, :, None]
        elif w.ndim == 3:  # Conv1d with k taps on a one-row image: the taps are the middle row of a k x k filter whose
            k = w.shape[2]  # other rows only ever multiply padding (spatial = 1 UNets, azula/nn/layers.py:25-50)
            w2 = torch.zeros(w.shape[0], w.shape[1], k, k, dtype=w.dtype, device=w.device)
            w2[:, :, k // 2, :] = w
            w = w2
        if w.shape[2] != w.shape[3]:  # anisotropic odd kernel (kh, kw): centred in a square one; with 'same' padding the
            kh, kw_ = w.shape[2], w.shape[3]  # extra zero rows / columns only ever add zeros
            assert kh % 2 == 1 and kw_ % 2 == 1, "odd kernel sizes only"
            k = max(kh, kw_)
            w2 = torch.zeros(w.shape[0], w.shape[1], k, k, dtype=w.dtype, device=w.device)
            w2[:, :, (k - kh) // 2 : (k - kh) // 2 + kh, (k - kw_) // 2 : (k - kw_) // 2 + kw_] = w
            w = w2
        self.w = w.contiguous()
        self.cout, self.cin, self.ks, kw = self.w.shape
        assert self.ks == kw
        self.cin0 = self.cin if cin0 is None else cin0
        self.c0s, self.c1s = bld.pad(self.cin0), bld.pad(self.cin - self.cin0)
        self.cout_s = bld.pad(self.cout)
        self.device = bld.device
        self.bias = None
        if bias is not None:
            self.bias = torch.zeros(self.cout_s, dtype=torch.float32, device=bld.device)
            self.bias[: self.cout] = bias.detach().to(device=bld.device, dtype=torch.float32)
        self._direct = self._wino = self._wino4 = self._x3 = self._h2 = self._wino_h2 = None
        self._half: dict = {}
        self._amax = None

    def direct(self) -> torch.Tensor:
        r"""[tap][cout_s][cin_s] (K contiguous), zero padded (az_pack_conv_weight_f32)."""
        if self._direct is None:
            cin_s = self.c0s + self.c1s
            packed = torch.empty(self.ks * self.ks * self.cout_s * cin_s, dtype=torch.float32, device=self.device)
            _lib.call(
                "az_pack_conv_weight_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.ks,
                self.cout_s, self.cin0, self.c0s, cin_s, _lib.stream_ptr(),
            )
            self._direct = packed
        return self._direct

    def stem(self) -> torch.Tensor:
        r"""(3, 3, cin, cout_s): tap-major, output channels contiguous (``az_conv2d_stem_f32``)."""
        if getattr(self, "_stem", None) is None:
            w = torch.zeros(self.ks, self.ks, self.cin, self.cout_s, dtype=torch.float32, device=self.device)
            w[..., : self.cout] = self.w.permute(2, 3, 1, 0)
            self._stem = w.contiguous()
        return self._stem

    def direct_half(self, f16: bool) -> torch.Tensor:
        r"""The direct layout in bf16 (``f16=False``) or IEEE half, for ``az_conv2d_{bf16,f16}_f32``."""
        if f16 not in self._half:
            cin_s = self.c0s + self.c1s
            packed = torch.empty(self.ks * self.ks * self.cout_s * cin_s, dtype=torch.int16, device=self.device)
            _lib.call(
                "az_pack_conv_weight_half_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.ks,
                self.cout_s, self.cin0, self.c0s, cin_s, int(f16), _lib.stream_ptr(),
            )
            self._half[f16] = packed
        return self._half[f16]

    def direct_x3(self) -> torch.Tensor:
        r"""The direct layout as three bf16 planes (w = w1 + w2 + w3 exactly), for ``az_conv2d_x3_f32``."""
        if self._x3 is None:
            cin_s = self.c0s + self.c1s
            packed = torch.empty(3 * self.ks * self.ks * self.cout_s * cin_s, dtype=torch.int16, device=self.device)
            _lib.call(
                "az_pack_conv_weight_x3_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.ks,
                self.cout_s, self.cin0, self.c0s, cin_s, _lib.stream_ptr(),
            )
            self._x3 = packed
        return self._x3

    def w_scale(self, winograd: bool) -> float:
        r"""The power of two the f16x2 packings multiply the weights by (``az_f16x2_weight_scale``: the largest magnitude -- of
        the Winograd-domain filter when ``winograd`` -- lands in [2^13, 2^14)); handed to the kernels as ``AzConvArgs.w_scale``."""
        if self._amax is None:
            self._amax = float(self.w.abs().max()) if self.w.numel() else 0.0
        return float(_lib.lib().az_f16x2_weight_scale(self._amax, int(winograd)))

    def direct_f16x2(self) -> torch.Tensor:
        r"""The direct layout as three IEEE half planes [wh | wl | wh / 2^11] of w * w_scale, for ``az_conv2d_f16x2_f32``."""
        if self._h2 is None:
            cin_s = self.c0s + self.c1s
            packed = torch.empty(3 * self.ks * self.ks * self.cout_s * cin_s, dtype=torch.int16, device=self.device)
            _lib.call(
                "az_pack_conv_weight_f16x2_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.ks,
                self.cout_s, self.cin0, self.c0s, cin_s, self.w_scale(False), _lib.stream_ptr(),
            )
            self._h2 = packed
        return self._h2

    def winograd_f16x2(self) -> torch.Tensor:
        r"""The x3 Winograd filter layout with the f16x2 pieces of U * w_scale, for ``az_conv2d_winograd_f16x2_f32``."""
        if self._wino_h2 is None:
            nk0, nk1 = (self.c0s + 15) // 16, (self.c1s + 15) // 16
            cb = (self.cout_s + 63) // 64
            packed = torch.empty((nk0 + nk1) * cb * 16 * 64 * 16 * 3, dtype=torch.int16, device=self.device)
            _lib.call(
                "az_winograd_pack_filter_f16x2_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.cin0,
                nk0, nk0 + nk1, cb, self.w_scale(True), _lib.stream_ptr(),
            )
            self._wino_h2 = packed
        return self._wino_h2

    def winograd(self) -> torch.Tensor:
        r"""Filter transform U = G g G^T (az_winograd_pack_filter_f32: fp64 accumulate, one-off) laid out
        [8-channel chunk][64-cout block][16 frequencies][64][8]; source 1 starts on a chunk boundary."""
        if self._wino is None:
            nk0, nk1 = (self.c0s + 7) // 8, (self.c1s + 7) // 8
            cb = (self.cout_s + 63) // 64
            packed = torch.empty((nk0 + nk1) * cb * 16 * 64 * 8, dtype=torch.float32, device=self.device)
            _lib.call(
                "az_winograd_pack_filter_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.cin0,
                nk0, nk0 + nk1, cb, _lib.stream_ptr(),
            )
            self._wino = packed
        return self._wino


    def winograd_x3(self) -> torch.Tensor:
        r"""The same filter transform as three bf16 pieces in MFMA fragment order, 16-channel steps
        (az_winograd_pack_filter_x3_f32), for ``az_conv2d_winograd_x3_f32``; source 1 starts on a step boundary."""
        if getattr(self, "_wino_x3", None) is None:
            nk0, nk1 = (self.c0s + 15) // 16, (self.c1s + 15) // 16
            cb = (self.cout_s + 63) // 64
            packed = torch.empty((nk0 + nk1) * cb * 16 * 64 * 16 * 3, dtype=torch.int16, device=self.device)
            _lib.call(
                "az_winograd_pack_filter_x3_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.cin0,
                nk0, nk0 + nk1, cb, _lib.stream_ptr(),
            )
            self._wino_x3 = packed
        return self._wino_x3

    def winograd4(self) -> torch.Tensor:
        r"""F(4x4,3x3) filter transform (az_winograd4_pack_filter_f32) laid out
        [4-channel chunk][64-cout block][36 frequencies][64][4]."""
        if self._wino4 is None:
            nk = (self.c0s + self.c1s) // 4
            cb = (self.cout_s + 63) // 64
            packed = torch.empty(nk * cb * 36 * 64 * 4, dtype=torch.float32, device=self.device)
            _lib.call(
                "az_winograd4_pack_filter_f32", packed.data_ptr(), self.w.data_ptr(), self.cout, self.cin, self.cin0,
                self.c0s, nk, cb, _lib.stream_ptr(),
            )
            self._wino4 = packed
        return self._wino4


class Builder:
    r"""Emits kernels onto a tape; owns the pool, packed weights and the split-K workspace."""

    def __init__(self, device: torch.device, half: torch.dtype | None = None, half_act: bool = False) -> None:
        r"""``half``: torch.bfloat16 / torch.float16 routes every conv / token GEMM through the half-operand MFMA
        kernel (fp32 accumulate) -- set by the plans of modules cast to half precision.  ``half_act``: the plan also keeps its
        activations in HBM in that type (HALF_ACT; channel strides are then multiples of 8)."""
        self.device = device
        self.half = half if half in (torch.bfloat16, torch.float16) else None
        self.half_act = bool(half_act) and self.half is not None and HALF_ACT
        self.tape = Tape()
        self.pool = Pool(device)
        self._ws_need = 0
        self._ws_users: list[AzConvArgs] = []
        self.workspace: torch.Tensor | None = None

    # -- buffers ---------------------------------------------------------------------------
    def pad(self, c: int) -> int:
        r"""Channel stride of this plan's activations: multiples of 4 floats, or of 8 two-byte values (16-byte vectors either way)."""
        return pad8(c) if self.half_act else pad4(c)

    def new_act(self, B: int, H: int, W: int, C_: int, pinned: bool = False, f32: bool = False) -> Act:
        r"""``f32``: an fp32 tensor also in a half-activation plan (the plan's input / output tensors)."""
        cs = self.pad(C_)
        dtype = self.half if (self.half_act and not f32) else torch.float32
        return Act(self.pool.alloc(B * H * W * cs, dtype), B, H, W, C_, cs, pinned)

    def free(self, a: Act) -> None:
        if not a.pinned:
            self.pool.release(a.buf)

    def const(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.tape.keep.append(t)
        return t

    def empty(self, *shape) -> torch.Tensor:
        t = torch.empty(*shape, dtype=torch.float32, device=self.device)
        self.tape.keep.append(t)
        return t

    # -- weights ---------------------------------------------------------------------------
    def pack_conv(self, weight: torch.Tensor, bias: torch.Tensor | None, cin0: int | None = None) -> "ConvWeights":
        r"""Wraps torch (cout, cin, k, k) [or (cout, cin) / (cout, cin, 1)] weights; the kernel-specific
        packed forms (direct implicit GEMM, Winograd) are materialised on first use."""
        cw = ConvWeights(self, weight, bias, cin0)
        self.tape.keep.append(cw)
        return cw

    # -- kernels ---------------------------------------------------------------------------
    def conv(
        self,
        src0: Act,
        packed,
        cout: int,
        *,
        src1: Act | None = None,
        up0: int = 0,
        up1: int = 0,
        hin: int | None = None,
        win: int | None = None,
        stride: int | tuple = 1,
        act: int = 0,
        gate: torch.Tensor | None = None,
        gate_off: int = 0,
        gate_bstride: int = 0,
        res: Act | None = None,
        res_up: int = 0,
        dst_nchw: torch.Tensor | None = None,
        winograd: bool | int | None = None,
        periodic: bool = False,
        gn_stats: bool = False,
        out: Act | None = None,
        depth: tuple | None = None,
        qk_prep: dict | None = None,
        out_f32: bool = False,
    ) -> Act | None:
        ks, bias = packed.ks, packed.bias
        pad = ks // 2
        B = src0.B
        # (height, width) pairs select the anisotropic descriptor (a stride sequence such as (2, 1), unet.py:159-186)
        (stride, stride_w), (up0, up0_w), (up1, up1_w) = (v if isinstance(v, (tuple, list)) else (v, v) for v in (stride, up0, up1))
        aniso = stride != stride_w or up0 != up0_w or up1 != up1_w
        if hin is None:
            hin = src0.H << up0
        if win is None:
            win = src0.W << up0_w
        hout = (hin + 2 * pad - ks) // stride + 1
        wout = (win + 2 * pad - ks) // stride_w + 1
        a = AzConvArgs()
        a.src0, a.c0s, a.up0, a.h0, a.w0 = src0.ptr, src0.cs, up0, src0.H, src0.W
        if aniso:
            a.aniso, a.stride_w, a.up0_w, a.up1_w = 1, stride_w, up0_w, up1_w
        if src1 is not None:
            a.src1, a.c1s, a.up1, a.h1, a.w1 = src1.ptr, src1.cs, up1, src1.H, src1.W
        assert (a.c0s, a.c1s) == (packed.c0s, packed.c1s), "weights were packed for different source strides"
        a.batch, a.hin, a.win = B, hin, win
        if depth is not None:  # (planes per volume, depth tap offset[, circular]): one depth tap of a 3-D convolution over all planes at once
            assert (src1 is None or src1.B == B) and B % depth[0] == 0
            a.depth, a.depth_shift = depth[0], depth[1]
            a.depth_wrap = int(bool(depth[2])) if len(depth) > 2 else 0
        a.bias = bias.data_ptr() if bias is not None else None
        a.cout_s = self.pad(cout)
        if self.half_act:  # typed tensors (AzConvArgs.src_dtype / dst_dtype): sources as they are, the destination in the module's type
            assert src1 is None or src1.half == src0.half, "both sources in one element type"
            a.src_dtype = int(src0.half)
        a.ksize, a.stride, a.pad = ks, stride, pad
        a.pad_mode = 1 if (periodic and pad > 0) else 0
        a.hout, a.wout = hout, wout
        a.act = act
        if gate is not None:
            a.gate = gate.data_ptr() + 4 * gate_off
            a.gate_bstride = gate_bstride
        if res is not None:
            if hasattr(res, "act"):  # batch-shared residual (positional embedding table)
                res = res.act
                a.res_bcast = 1
            a.res, a.res_up, a.hres, a.wres = res.ptr, res_up, res.H, res.W
            assert res.cs == a.cout_s
        if dst_nchw is not None:
            out = None
            a.dst, a.dst_nchw, a.dst_c = dst_nchw.data_ptr(), 1, cout
        else:
            if out is None:
                if act == 4:  # SwiGLU epilogue: half the channels come out (y[c] = x[2c] * silu(x[2c+1]))
                    assert cout % 8 == 0 and gate is None and res is None, "SwiGLU epilogue: cout % 8 == 0, no gate / residual"
                    assert not self.half_act or cout % 16 == 0
                    out = self.new_act(B, hout, wout, cout // 2, f32=out_f32)
                else:
                    out = self.new_act(B, hout, wout, cout, f32=out_f32)
            else:  # caller-owned destination (a plane range of a volume; may alias `res`: in-place accumulation)
                assert (out.B, out.H, out.W, out.C, out.cs) == (B, hout, wout, cout, self.pad(cout)), "destination shape"
            a.dst = out.ptr
            a.dst_dtype = int(out.half)
            if res is not None:
                assert res.half == out.half, "the residual has the destination's element type"
        npix = B * hout * wout
        cin_s = a.c0s + a.c1s
        lib = _lib.lib()
        # (half-precision modules: the direct bf16 / f16 kernel; one factor per axis: the direct kernel's loaders only)
        legal = ks == 3 and stride == 1 and not aniso and self.half is None
        tiles4 = B * ((hout + 3) // 4) * ((wout + 3) // 4)
        head_wgs = B * ((hout + 15) // 16) * ((wout + 15) // 16)
        # image head (<= 4 output channels) on a map that fills the chip with 16 x 16-pixel workgroups:
        # az_conv2d_f32 runs its narrow-output VALU kernel (104 vs 342 us at 4 x 256^2, 256 -> 3)
        head = (winograd is None and legal and not aniso and a.cout_s == 4 and src1 is None and up0 == 0 and a.c0s % 16 == 0
                and head_wgs >= 256 and depth is None)
        wino_ok = legal and not head and winograd not in ("x3", "h2", "h2d")
        use_f4 = wino_ok and depth is None and (winograd == 4 or (winograd is None and WINOGRAD == "4" and tiles4 >= WINOGRAD4_MIN_TILES))
        use_wino = wino_ok and not use_f4 and ((WINOGRAD != "0") if winograd is None else bool(winograd))
        if use_wino:
            tiles = B * ((hout + 1) // 2) * ((wout + 1) // 2)
            a.splitk = lib.az_conv2d_winograd_suggest_splitk(B, hout, wout, a.cout_s, cin_s)
            if winograd is None and WINOGRAD in ("1", "4") and tiles < 64:
                use_wino = False  # less than one 64-tile block (measured: 8x8 at batch < 4): the direct kernel wins
        # bf16x3 mode replaces the DIRECT fp32 kernel (1x1 convs / token GEMMs, stride 2, small maps: 147-181 vs 113-128
        # TF/s); the 3x3 stride-1 layers stay on the fp32 Winograd kernel, which executes 2.25x fewer multiplies
        # (221 vs 181 TF/s algorithmic at 4 x 256^2, 256 -> 256).
        use_x3 = self.half is None and (
            winograd in ("x3", "h2", "h2d")
            or (winograd is None and pieces() and not head and not use_wino and not use_f4
                and cin_s >= X3_MIN_CHANNELS and a.cout_s >= X3_MIN_CHANNELS)
        )
        # (winograd = "x3" / "wx3": the bf16x3 kernels, "h2" / "wh2": the f16x2 ones, "h2d" / "wh2d": those with the activation scale
        #  measured from the sources -- az_absmax_f32 --, whatever the mode: kernel tests)
        src_bounded = (src0.bounded or src0.affine is not None) and (src1 is None or src1.bounded)  # (a pending normalisation is applied in the gather / materialised)
        h2 = winograd in ("h2", "wh2", "h2d", "wh2d") or (winograd not in ("x3", "wx3") and FP32_MFMA == "f16x2" and src_bounded)
        dyn = winograd in ("h2d", "wh2d")
        if (not h2 and winograd is None and FP32_MFMA == "f16x2" and F16X2_DYNAMIC and self.half is None and depth is None
                and (use_wino or use_x3) and not use_f4):
            # unbounded sources: f16x2 with the scale measured per step, where the pass over the sources costs clearly less than the
            # matrix instructions it saves (~12 % of a Winograd layer at ~350 TF/s algorithmic, ~25 % of a direct one at ~190)
            flops = 2.0 * npix * cout * (src0.C + (src1.C if src1 is not None else 0)) * ks * ks
            gain_s = 0.12 * flops / 350e12 if use_wino else 0.25 * flops / 190e12
            cost_s = sum(4e-6 + (0.0 if (s_.gn_quads is not None and F16X2_MOMENTS) else s_.buf.numel() * 4 / 5.0e12)
                         for s_ in (src0, src1) if s_ is not None and s_.absmax is None)
            dyn = h2 = gain_s > 1.5 * cost_s
        if use_f4:
            a.weight = packed.winograd4().data_ptr()
            a.splitk = lib.az_conv2d_winograd4_suggest_splitk(B, hout, wout, a.cout_s, cin_s)
            name = "az_conv2d_winograd4_f32"
        elif use_wino and wout >= 3 and (winograd in ("wx3", "wh2", "wh2d") or (winograd is None and WINO_X3 and pieces())):
            # the frequency GEMMs on the bf16 pipe as exact 3 x bf16 splits (wino_x3.hip); same descriptor, 16-channel steps
            # (f16x2: the same kernel with two half pieces per operand and three products)
            if h2:
                a.weight, a.w_scale = packed.winograd_f16x2().data_ptr(), packed.w_scale(True)
                name = "az_conv2d_winograd_f16x2_f32"
                if dyn:
                    a.in_absmax0 = self.absmax_of(src0).data_ptr()
                    a.in_absmax1 = self.absmax_of(src1).data_ptr() if src1 is not None else None
            else:
                a.weight = packed.winograd_x3().data_ptr()
                name = "az_conv2d_winograd_x3_f32"
        elif use_wino:
            a.weight = packed.winograd().data_ptr()
            name = "az_conv2d_winograd_f32"
        elif use_x3:
            if h2:
                a.weight, a.w_scale = packed.direct_f16x2().data_ptr(), packed.w_scale(False)
                if dyn:
                    a.in_absmax0 = self.absmax_of(src0).data_ptr()
                    a.in_absmax1 = self.absmax_of(src1).data_ptr() if src1 is not None else None
            else:
                a.weight = packed.direct_x3().data_ptr()
            a.splitk = lib.az_conv2d_x3_suggest_splitk(C.byref(a))  # (the 256 x 256-tile kernel has its own rule)
            name = "az_conv2d_f16x2_f32" if h2 else "az_conv2d_x3_f32"
        elif self.half is not None:
            a.weight = packed.direct_half(self.half == torch.float16).data_ptr()
            a.splitk = lib.az_conv2d_suggest_splitk(npix, a.cout_s, cin_s, ks)
            if a.c0s % 64 == 0 and a.c1s % 64 == 0:  # (the 256 x 256-tile kernel has its own rule where it takes the launch)
                a.splitk = lib.az_conv2d_x3_suggest_splitk(C.byref(a))
            name = "az_conv2d_f16_f32" if self.half == torch.float16 else "az_conv2d_bf16_f32"
        else:
            a.weight = packed.direct().data_ptr()
            a.splitk = lib.az_conv2d_suggest_splitk(npix, a.cout_s, cin_s, ks)
            name = "az_conv2d_f32"
        tmp_src = None
        if src0.affine is not None:  # a normalisation whose apply pass has not run (group_norm(lazy=True))
            ST, in_act = src0.affine
            # the apply pass inside the gather (the fp32 kernel on a source of the output's size only; the x3 kernel's patch masks
            # live in output coordinates, so it also reads a nearest-upsampled source)
            if src1 is None and a.c0s % 8 == 0 and not aniso and (
                    name in WINO_X3_NAMES or (name == "az_conv2d_winograd_f32" and up0 == 0)):
                a.in_affine, a.in_act = ST.data_ptr(), in_act
            else:
                tmp_src = self.materialize(src0)
                a.src0 = tmp_src.ptr
        if (gn_stats and GN_FUSED and name in ("az_conv2d_winograd_f32", *WINO_X3_NAMES) and a.splitk == 1 and out is not None and cout == a.cout_s
                and cout % 64 == 0 and hout % 2 == 0 and wout % 2 == 0 and ((hout // 2) * (wout // 2)) % 64 == 0):
            # the output feeds a GroupNorm: its epilogue also writes per-(tile block, channel quad) moments
            chunks = ((hout // 2) * (wout // 2)) // 64
            quads = torch.empty(B * chunks * (cout // 4) * 4, dtype=torch.float32, device=self.device)
            a.gn_quads, a.gn_chunks = quads.data_ptr(), chunks
            out.gn_quads = (quads, chunks)
            self.tape.keep.append(quads)
        elif (gn_stats and GN_FUSED and GN_FUSED_SPLITK and a.splitk > 1 and out is not None and cout == a.cout_s and name != "az_conv2d_winograd4_f32"
              and not (name == "az_conv2d_f32" and a.cout_s == 4)):
            # split-K layers (the small maps): the combine kernel leaves the moments, one partial per (image, pixel chunk, quad)
            hw = hout * wout
            # a workgroup of the combine kernel = 256 // quads pixel slots; about 2 pixels per thread (each costs splitk
            # dependent-latency loads: parallelism, not bandwidth, decides), at most 128 partials per image (two finalize passes)
            cpix = 2 * max(1, 256 // (cout // 4))
            chunks = max(1, min(128, (hw + cpix - 1) // cpix))
            while (hw + chunks - 1) // chunks * (chunks - 1) >= hw:
                chunks -= 1
            quads = torch.empty(B * chunks * (cout // 4) * 4, dtype=torch.float32, device=self.device)
            a.gn_quads, a.gn_chunks = quads.data_ptr(), chunks
            out.gn_quads = (quads, chunks)
            self.tape.keep.append(quads)
        if (qk_prep is not None and QK_PREP and name in ("az_conv2d_f32", "az_conv2d_bf16_f32", "az_conv2d_f16_f32", "az_conv2d_x3_f32", "az_conv2d_f16x2_f32")
                and a.splitk == 1 and act == 0 and gate is None and res is None and out is not None and a.cout_s == cout
                and qk_prep["head_dim"] in (32, 64, 128) and cout == 3 * qk_prep["heads"] * qk_prep["head_dim"]):
            # the fused q | k | v projection of an attention layer: q / k RMS norm, gains and RoPE in THIS epilogue, once per
            # layer, instead of in every workgroup of the attention kernel (three per head at 288 tokens: 123 -> 163 us)
            a.act = 5
            if name != "az_conv2d_f32" and lib.az_conv2d_x3_suggest_splitk(C.byref(a)) != 1:
                a.act = 0  # (the tile plan of THIS epilogue wants a split K walk, which the epilogue cannot take: plain projection)
        if a.act == 5:
            a.qk_head_dim, a.qk_heads, a.qk_tokens = qk_prep["head_dim"], qk_prep["heads"], hout * wout
            a.qk_rmsnorm, a.qk_eps = int(qk_prep["rmsnorm"]), qk_prep["eps"]
            keep = []
            if qk_prep.get("weight") is not None:
                a.qk_q_weight, a.qk_k_weight = (t.data_ptr() for t in qk_prep["weight"])
                keep += list(qk_prep["weight"])
            if qk_prep.get("rope") is not None:
                a.qk_rope_cos, a.qk_rope_sin = (t.data_ptr() for t in qk_prep["rope"])
                keep += list(qk_prep["rope"])
            self.tape.keep.extend(keep)
            out.qk_prepared = True
        if a.splitk > 1:
            self._ws_need = max(self._ws_need, a.splitk * npix * a.cout_s)
            self._ws_users.append(a)
        a._flops = 2 * npix * cout * (src0.C + (src1.C if src1 is not None else 0)) * ks * ks  # algorithmic
        a._algo = name
        self.tape.add(name, C.byref(a), keep=[a] if gate is None else [gate, a])  # the descriptor holds raw addresses
        if tmp_src is not None:
            self.free(tmp_src)
        if out is not None:
            out.absmax = None  # (a caller-owned destination is rewritten: maxima recorded for its previous contents are stale)
            out.bounded = src_bounded and (res is None or res.bounded)  # (a residual add of the stream joins the stream; of a bounded tensor -- y + MSA(y) of a DiT block -- stays bounded)
        return out

    def conv_stem(self, x: torch.Tensor, B: int, cin: int, H: int, W: int, packed: "ConvWeights", cout: int, *,
                  periodic: bool = False, gn_stats: bool = False) -> Act:
        r"""The network's first 3 x 3 convolution reading its <= 4 input channels PLANAR -- ``x`` is the (B, cin, H, W) latent
        as the sampler holds it (``az_conv2d_stem_f32``): no NHWC copy of the latent, no matrix kernel for 27 multiplies."""
        assert packed.ks == 3 and 1 <= cin <= 4 and packed.cin == cin and cout % 4 == 0 and self.half is None
        a = AzConvArgs()
        a.src0, a.c0s, a.h0, a.w0 = x.data_ptr(), cin, H, W
        a.batch, a.hin, a.win, a.hout, a.wout = B, H, W, H, W
        a.weight = packed.stem().data_ptr()
        a.bias = packed.bias.data_ptr() if packed.bias is not None else None
        a.cout_s, a.ksize, a.stride, a.pad, a.splitk = cout, 3, 1, 1, 1
        a.pad_mode = 1 if periodic else 0
        out = self.new_act(B, H, W, cout)
        a.dst = out.ptr
        if gn_stats and GN_FUSED:
            chunks = ((H + 7) // 8) * ((W + 31) // 32)
            quads = torch.empty(B * chunks * (cout // 4) * 4, dtype=torch.float32, device=self.device)
            a.gn_quads, a.gn_chunks = quads.data_ptr(), chunks
            out.gn_quads = (quads, chunks)
            self.tape.keep.append(quads)
        a._flops = 0  # (27 multiplies per output on the vector ALUs: a store-bound pass, accounted by its bytes in bench.py)
        a._algo = "az_conv2d_stem_f32"
        self.tape.add("az_conv2d_stem_f32", C.byref(a), keep=[a, x])
        return out

    def absmax_of(self, x: Act) -> torch.Tensor:
        r"""The AZ_ABSMAX_SLOTS partial maxima of |x| (``az_absmax_f32``: one streaming pass, recorded once per tensor and shared by
        its consumers) -- the activation scale of an f16x2 launch on an unbounded input (``AzConvArgs.in_absmax0 / in_absmax1``).
        Valid as long as the tensor is not rewritten on the tape: ops that write into an existing Act reset ``absmax`` (``conv(out=...)``)."""
        if x.absmax is None:
            assert not x.half and x.affine is None
            slots = self.empty(256)
            if x.gn_quads is not None and F16X2_MOMENTS:
                # the producing convolution left GroupNorm moments of this tensor: |x| <= |mean| + sqrt(M2) per record -- an upper
                # bound of the maximum for the price of reading the records (az_absmax_from_moments_f32)
                q = x.gn_quads[0]
                self.tape.add("az_absmax_from_moments_f32", slots.data_ptr(), q.data_ptr(), q.numel() // 4, keep=[q])
            else:
                self.tape.add("az_absmax_f32", slots.data_ptr(), x.ptr, x.B * x.H * x.W * x.cs, keep=[x.buf])
            x.absmax = slots
        return x.absmax

    def upsample_nearest(self, x: Act, sh: int, sw: int, hout: int, wout: int) -> Act:
        r"""``narrow(Upsample(scale_factor=(sh, sw), mode="nearest")(x), (hout, wout))`` as a pass of its own -- only for
        factors that are not powers of two (those are a shift inside the consuming convolution's gather)."""
        y = self.new_act(x.B, hout, wout, x.C)
        y.bounded = x.bounded
        self.tape.add("az_upsample_nearest_f32", y.ptr, x.ptr, x.B, x.H, x.W, x.cs, sh, sw, hout, wout)
        return y

    def finish(self) -> None:
        r"""Allocates the shared split-K workspace and patches it into the recorded convs."""
        if self._ws_need and (self.workspace is None or self.workspace.numel() < self._ws_need):
            self.workspace = torch.empty(self._ws_need, dtype=torch.float32, device=self.device)
        for a in self._ws_users:
            a.workspace = self.workspace.data_ptr()
        self._ws_users = []

    def linear_small(self, y, ldy, x, ldx, W, bias, M, N, K, in_act=0, out_act=0, y_off=0) -> None:
        self.tape.add(
            "az_linear_small_f32", y.data_ptr() + 4 * y_off, ldy, x.data_ptr(), ldx, W.data_ptr(),
            bias.data_ptr() if bias is not None else None, M, N, K, in_act, out_act, keep=[y, x, W, bias],
        )

    def materialize(self, x: Act) -> Act:
        r"""The apply pass of a lazy normalisation (``Act.affine``) as a tensor of its own."""
        ST, act = x.affine
        n = x.B * x.cs
        y = self.new_act(x.B, x.H, x.W, x.C, f32=not x.half)
        y.bounded = True
        self._affine_act(y, x, None, 0, ST.data_ptr(), ST.data_ptr() + 4 * n, x.B, x.H, x.W, x.cs, act, 0)
        return y

    def _affine_act(self, y: Act, x: Act, x1p, c0s: int, S: int, T: int, B: int, H: int, W: int, cs: int, act: int, pool: int) -> None:
        r"""y = act(x * S + T) (optionally pooled) on fp32 tensors or on tensors in the module's 2-byte type (x, x1, y alike)."""
        if x.half:
            assert y.half
            self.tape.add("az_affine_act_h16", y.ptr, x.ptr, x1p, c0s, S, T, B, H, W, cs, act, pool, 2 if x.buf.dtype == torch.float16 else 1)
        else:
            self.tape.add("az_affine_act_f32", y.ptr, x.ptr, x1p, c0s, S, T, B, H, W, cs, act, pool)

    def group_norm(
        self, x: Act, groups: int, *, weight=None, bias=None, scale=None, shift=None, scale_off=0, shift_off=0,
        bstride=0, act=0, pool=0, eps=1e-5, x1: Act | None = None, lazy: bool = False,
    ) -> Act:
        r"""y = act((GN(x)*w + b) * (1 + scale) + shift), optionally 2x2 average pooled.  With ``x1`` the
        input is the channel concatenation [x | x1], read in place (never materialised)."""
        B, HW = x.B, x.H * x.W
        x1p, c0s = None, 0
        src_quads = [x.gn_quads] + ([x1.gn_quads] if x1 is not None else [])
        src_channels = [x.C] + ([x1.C] if x1 is not None else [])
        if x1 is not None:
            assert x.C == x.cs and x1.C == x1.cs and (x1.H, x1.W) == (x.H, x.W) and x1.half == x.half
            x1p, c0s = x1.ptr, x.cs
            x = Act(x.buf, x.B, x.H, x.W, x.C + x1.C, x.cs + x1.cs, True)
        ST = self.empty(2 * B * x.cs)  # [scale | shift]: one buffer (AzConvArgs.in_affine reads both through one descriptor)
        S, T = ST[: B * x.cs], ST[B * x.cs :]
        f = AzNormFinalizeArgs()
        Cg = x.C // groups
        fused = src_quads is not None and Cg % 4 == 0 and x.C == x.cs and all(q is not None for q in src_quads) \
            and all((c // 4) % (Cg // 4) == 0 for c in src_channels)
        if fused:  # every source was produced by a convolution that left its moments: no statistics pass
            nchunks = src_quads[0][1]
            f.partials = src_quads[0][0].data_ptr()
            if len(src_quads) > 1:
                f.partials1, f.nchunks1 = src_quads[1][0].data_ptr(), src_quads[1][1]
            f.quads_per_group, f.quads0 = Cg // 4, src_channels[0] // 4
        else:
            nchunks = int(min(512, max(1, (HW * x.cs * 4) // 65536)))  # ~64 KB of x per workgroup
            partials = self.empty(B * nchunks * groups * 4)
            if x.half:
                self.tape.add("az_groupnorm_stats_h16", partials.data_ptr(), x.ptr, x1p, c0s, B, HW, x.C, x.cs, groups, nchunks,
                              2 if x.buf.dtype == torch.float16 else 1)
            else:
                self.tape.add("az_groupnorm_stats_f32", partials.data_ptr(), x.ptr, x1p, c0s, B, HW, x.C, x.cs, groups, nchunks)
            f.partials = partials.data_ptr()
        f.S, f.T = S.data_ptr(), T.data_ptr()
        f.weight = weight.data_ptr() if weight is not None else None
        f.bias = bias.data_ptr() if bias is not None else None
        f.scale = scale.data_ptr() + 4 * scale_off if scale is not None else None
        f.shift = shift.data_ptr() + 4 * shift_off if shift is not None else None
        f.scale_bstride = bstride
        f.B, f.C, f.cs, f.groups, f.nchunks, f.eps = B, x.C, x.cs, groups, nchunks, eps
        # (the descriptor holds raw addresses: the tensors behind them must live as long as the tape)
        self.tape.add("az_groupnorm_finalize_f32", C.byref(f), keep=[f, weight, bias, scale, shift])
        if (lazy and AFFINE_FUSED and x1 is None and not pool and act == 0 and x.C == x.cs and x.cs % 8 == 0
                and self.half is None):
            # no apply pass: the consumer (Builder.conv) reads x and applies scale / shift itself
            y = Act(x.buf, B, x.H, x.W, x.C, x.cs, True)
            y.affine = (ST, act)
            y.bounded = True  # (the values the consumer sees: act(buf * scale + shift))
            return y
        if pool:  # 1: 2x2, 2: along the width only (a 1-D signal held as a one-row image)
            y = self.new_act(B, x.H // 2 if pool == 1 else x.H, x.W // 2, x.C, f32=not x.half)
        else:
            y = self.new_act(B, x.H, x.W, x.C, f32=not x.half)
        self._affine_act(y, x, x1p, c0s, S.data_ptr(), T.data_ptr(), B, x.H, x.W, x.cs, act, pool)
        y.bounded = True
        return y

    def row_norm(self, x: Act, kind: int, *, weight=None, scale=None, shift=None, scale_off=0, shift_off=0, bstride=0,
                 eps=1e-5):
        y = self.new_act(x.B, x.H, x.W, x.C, f32=not x.half)
        y.bounded = True
        rows = x.B * x.H * x.W
        args = (y.ptr, x.ptr, weight.data_ptr() if weight is not None else None,
                scale.data_ptr() + 4 * scale_off if scale is not None else None,
                shift.data_ptr() + 4 * shift_off if shift is not None else None,
                bstride, rows, x.H * x.W, x.C, x.cs, kind, eps)
        if x.half:  # rows in the module's 2-byte type (fp32 statistics / modulation)
            self.tape.add("az_rownorm_mod_h16", *args, 2 if x.buf.dtype == torch.float16 else 1, keep=[weight, scale, shift])
        else:
            self.tape.add("az_rownorm_mod_f32", *args, keep=[weight, scale, shift])
        return y


# ------------------------------------------------------------------------------- AdaZero modulation helpers
def ada_zero_triple(bld: "Builder", ada_zero, channels: int, D: int, mod_rows: int, mod_jobs: list):
    r"""(abc buffer, batch stride) of one block's modulation triple (a, b, c), each padded to ``pad4(channels)``.
    ``ada_zero`` is the block's ``Linear(D, D) -> SiLU -> Linear(D, 3C)`` (then the MLP is queued on ``mod_jobs`` and
    emitted by :func:`mod_front_tape` for all blocks at once) or its raw ``(3, C, ...)`` parameter
    (reference ``azula/nn/unet.py:63-75``, ``azula/nn/dit.py:57-68``)."""
    cs, device = pad4(channels), bld.device
    if not isinstance(ada_zero, torch.nn.Parameter):
        abc = bld.empty(max(mod_rows, 1), 3 * cs)
        l0, l2 = ada_zero[0], ada_zero[2]
        w2 = torch.zeros(3 * cs, D, dtype=torch.float32, device=device)
        b2 = torch.zeros(3 * cs, dtype=torch.float32, device=device)
        for n in range(3):
            w2[n * cs : n * cs + channels] = l2.weight.detach()[n * channels : (n + 1) * channels]
            b2[n * cs : n * cs + channels] = l2.bias.detach()[n * channels : (n + 1) * channels]
        mod_jobs.append((l0, bld.const(w2), bld.const(b2), abc, 3 * cs))
        return abc, (3 * cs if mod_rows > 1 else 0)
    abc = torch.zeros(3 * cs, dtype=torch.float32, device=device)
    for n in range(3):
        abc[n * cs : n * cs + channels] = ada_zero.detach()[n].flatten()
    return bld.const(abc), 0


def mod_front_tape(bld: "Builder", mod_jobs: list, mod_buf: torch.Tensor, mod_rows: int, D: int) -> Tape:
    r"""h_i = silu(W0_i mod + b0_i) for ALL queued blocks as one GEMV, abc_i = W2_i h_i + b2_i as one grouped GEMV."""
    from ._lib import AzLinearGroup

    nj, rows = len(mod_jobs), max(mod_rows, 1)
    w0 = bld.const(torch.cat([j[0].weight.detach() for j in mod_jobs]))
    b0 = bld.const(torch.cat([j[0].bias.detach() for j in mod_jobs]))
    h_all = bld.empty(rows, nj * D)
    groups = (AzLinearGroup * nj)()
    for i, (_, w2, b2, abc, n_out) in enumerate(mod_jobs):
        g = groups[i]
        g.y, g.x, g.W, g.bias = abc.data_ptr(), h_all.data_ptr() + 4 * i * D, w2.data_ptr(), b2.data_ptr()
        g.ldy, g.ldx, g.N, g.K = n_out, nj * D, n_out, D
    gdev = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).to(bld.device)
    pre = Tape()
    pre.add("az_linear_small_f32", h_all.data_ptr(), nj * D, mod_buf.data_ptr(), D, w0.data_ptr(), b0.data_ptr(), rows, nj * D, D, 0, 1)
    pre.add("az_linear_small_grouped_f32", gdev.data_ptr(), nj, max(j[4] for j in mod_jobs), rows, 0, 0, keep=[gdev, w0, b0, h_all])
    return pre


def transition_args(**kw) -> AzTransitionArgs:
    a = AzTransitionArgs()
    for k, v in kw.items():
        setattr(a, k, v)
    return a



# ------------------------------------------------------------------------------- token-path helpers
# Head sizes.  The reference accepts any channels // attention_heads (azula/nn/attention.py:35-51; guided-diffusion any
# num_head_channels); the gfx950 attention kernels are instantiated for 16 / 32 / 64 / 80 / 128 (the fp32 kernel also for 8).  Any
# other size d <= 128 runs ZERO-PADDED to the next instantiated size d': the q | k | v projection is packed with d' - d zero rows
# (and zero bias) per head, so the padded channels of q, k and v are exact zeros -- they change neither q.k nor p.v --, the output
# projection with d' - d zero input columns per head; the scale stays 1 / sqrt(d), the q / k RMS norm averages over d
# (AzAttnArgs.norm_dim), padded RoPE pairs do not turn (theta = 0) and padded gains are 1.  (G24: 24, 48, 96.)
ATTN_HEAD_DIMS = (16, 32, 64, 80, 128)


def attn_padded_dim(d: int, half=None) -> int:
    if d == 8 and half is None:
        return 8
    for v in ATTN_HEAD_DIMS:
        if d <= v:
            return v
    raise NotImplementedError(f"attention head size {d}: the gfx950 attention kernels go up to 128 channels per head")


def pad_qkv_heads(w: torch.Tensor, b: torch.Tensor | None, heads: int, d: int, dp: int, order: str):
    r"""(3 heads d, Cin[, 1]) q | k | v projection weights (and bias) -> (3 heads dp, Cin): dp - d zero rows behind every head's d.
    ``order`` as in Builder.attention: "nHC" / "3HC" = rows (n, head, c), "H3C" = rows (head, n, c)."""
    w = w.detach().float().reshape(3 * heads * d, -1)
    lead = (3, heads) if order in ("nHC", "3HC") else (heads, 3)
    wp = w.new_zeros(*lead, dp, w.shape[1])
    wp[:, :, :d] = w.reshape(*lead, d, w.shape[1])
    bp = None
    if b is not None:
        bp = w.new_zeros(*lead, dp)
        bp[:, :, :d] = b.detach().float().reshape(*lead, d)
        bp = bp.reshape(-1)
    return wp.reshape(3 * heads * dp, w.shape[1]), bp


def pad_proj_heads(w: torch.Tensor, heads: int, d: int, dp: int) -> torch.Tensor:
    r"""(Cout, heads d[, 1]) output projection -> (Cout, heads dp): zero columns for the padded channels of every head."""
    w = w.detach().float().reshape(w.shape[0], heads, d)
    wp = w.new_zeros(w.shape[0], heads, dp)
    wp[:, :, :d] = w
    return wp.reshape(w.shape[0], heads * dp)


def pad_head_table(t: torch.Tensor, heads: int, n: int, n_pad: int, fill: float = 0.0) -> torch.Tensor:
    r"""(..., heads n) per-head table (RoPE angles with n = d / 2) -> (..., heads n_pad), padded with ``fill``."""
    lead = t.shape[:-1]
    tp = t.new_full((*lead, heads, n_pad), fill)
    tp[..., :n] = t.reshape(*lead, heads, n)
    return tp.reshape(*lead, heads * n_pad)


def _builder_attention(self, qkv: Act, heads: int, order: str, qk_rmsnorm: bool, scale: float, eps: float = 1e-5,
                       rope: tuple | None = None, qk_weight: tuple | None = None, mask: torch.Tensor | None = None,
                       norm_dim: int = 0) -> Act:
    r"""softmax(q k^T * scale) v over a fused-QKV token tensor (B, L, 1, 3*heads*dim).

    order: "nHC" = azula '(n H C)' (attention.py:90), "H3C" = ADM legacy (unet.py:338),
    "3HC" = ADM new order (unet.py:371).  Output (B, L, 1, heads*dim) laid out '(H C)'.
    ``norm_dim``: the real head size of zero-padded heads (see ATTN_HEAD_DIMS)."""
    from ._lib import AzAttnArgs

    Cq = qkv.C // 3
    dim = Cq // heads
    assert qkv.cs == qkv.C and dim * heads == Cq
    L = qkv.H * qkv.W
    out = self.new_act(qkv.B, qkv.H, qkv.W, Cq, f32=not qkv.half)
    out.bounded = qkv.bounded  # (a convex combination of the values)
    a = AzAttnArgs()
    a.io_dtype = int(qkv.half)  # (q, k, v, out in the module's 2-byte type: the bf16 / f16 entries only)
    es = 2 if qkv.half else 4
    base = qkv.ptr
    if order in ("nHC", "3HC"):
        offs, hs = (0, Cq, 2 * Cq), dim
    elif order == "H3C":
        offs, hs = (0, dim, 2 * dim), 3 * dim
    else:
        raise ValueError(order)
    a.q, a.k, a.v, a.out = base + es * offs[0], base + es * offs[1], base + es * offs[2], out.ptr
    a.batch, a.heads, a.tokens, a.head_dim = qkv.B, heads, L, dim
    for n in ("q", "k", "v"):
        setattr(a, n + "_bstride", L * qkv.cs)
        setattr(a, n + "_tstride", qkv.cs)
        setattr(a, n + "_hstride", hs)
    a.o_bstride, a.o_tstride, a.o_hstride = L * out.cs, out.cs, dim
    a.scale, a.qk_rmsnorm, a.eps, a.norm_dim = scale, int(qk_rmsnorm), eps, norm_dim
    if qkv.qk_prepared:  # the projection's epilogue has normalised / gained / rotated q and k already (Builder.conv(qk_prep=...))
        assert order in ("nHC", "3HC")
        a.qk_rmsnorm, rope, qk_weight = 0, None, None
    if rope is not None:  # (cos, sin) tables of shape (L, heads * dim / 2)
        a.rope_cos, a.rope_sin = rope[0].data_ptr(), rope[1].data_ptr()
        self.tape.keep.extend(rope)
    if qk_weight is not None:  # learned (dim,) gains of the q / k RMS norms
        a.q_weight, a.k_weight = qk_weight[0].data_ptr(), qk_weight[1].data_ptr()
        self.tape.keep.extend(qk_weight)
    if mask is not None:  # (L, L), (B | 1, 1 | H, L, L) boolean: True = attend (reference attention.py:97-104)
        m = mask
        if m.ndim == 2:
            m = m[None, None]
        if m.ndim != 4 or m.shape[-2:] != (L, L) or m.shape[0] not in (1, qkv.B) or m.shape[1] not in (1, heads):
            raise ValueError(f"attention mask of shape {tuple(mask.shape)} does not broadcast to ({qkv.B}, {heads}, {L}, {L})")
        m8 = (m != 0).to(device=self.device, dtype=torch.uint8).contiguous()
        a.mask = m8.data_ptr()
        a.mask_bstride = m8.stride(0) if m.shape[0] > 1 else 0
        a.mask_hstride = m8.stride(1) if m.shape[1] > 1 else 0
        self.tape.keep.append(m8)
    a._flops = 4 * qkv.B * heads * L * L * dim
    name = "az_attention_f32"
    if self.half is None and pieces() and ATTN_X3 and dim in (16, 32, 64, 80):
        # the two contractions as 3 x bf16 pieces / 6 partial products: fp32 accuracy, 0.375 x the pipe time (64 x 12 heads x 256
        # tokens x 64: 140 -> 111 us; head_dim 128 needs one wave per SIMD there and measured slower, 458 vs 516 us: fp32 kernel)
        name = "az_attention_f16x2_f32" if FP32_MFMA == "f16x2" and ATTN_H2 and qkv.bounded else "az_attention_x3_f32"
    if self.half is not None:  # module cast to half precision: contractions on the bf16 / f16 MFMA
        name = "az_attention_f16_f32" if self.half == torch.float16 else "az_attention_bf16_f32"
    self.tape.add(name, C.byref(a), keep=[a])
    return out


Builder.attention = _builder_attention
