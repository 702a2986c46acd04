r"""ctypes binding of ``libazula_amd.so`` -- the C ABI declared in ``include/azula_amd.h``.

The library is plain HIP/C (no torch types); Python hands it ``tensor.data_ptr()`` and the raw
``hipStream_t`` of torch's current stream.  There is NO fallback: every device-tensor code
path of :mod:`azula_amd` goes through :func:`lib`, which raises if the shared object is missing.
"""

from __future__ import annotations

import ctypes as C
import functools
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# AZULA_AMD_LIB: an alternative build of the same C ABI (kernel A/B experiments: tools/ablate.py)
LIB_PATH = os.environ.get("AZULA_AMD_LIB") or os.path.join(_HERE, "csrc", "libazula_amd.so")

c_f32p = C.c_void_p  # device pointers travel as integers
c_stream = C.c_void_p


class AzStepCoef(C.Structure):
    _fields_ = [
        ("c_in", C.c_float),
        ("c_skip", C.c_float),
        ("c_out", C.c_float),
        ("c_time", C.c_float),
        ("alpha_t", C.c_float),
        ("alpha_s", C.c_float),
        ("k_x", C.c_float),
        ("k_eps", C.c_float),
        ("c_in_next", C.c_float),
        ("clip_lo", C.c_float),
        ("clip_hi", C.c_float),
        ("guidance", C.c_float),
        ("time_index", C.c_int32),
        ("step", C.c_int32),
        ("pad", C.c_float * 2),
    ]


# column order of the float32 view of an AzStepCoef row (ints are bit-cast into their slots)
COEF_FIELDS = [f[0] for f in AzStepCoef._fields_[:14]]
COEF_WORDS = 16
assert C.sizeof(AzStepCoef) == 64


class AzTransitionArgs(C.Structure):
    _fields_ = [
        ("x_t", c_f32p),
        ("F", c_f32p),
        ("F_neg", c_f32p),
        ("eps", c_f32p),
        ("x_s", c_f32p),
        ("xin_next", c_f32p),
        ("mean_out", c_f32p),
        ("batch", C.c_int64),
        ("channels", C.c_int64),
        ("inner", C.c_int64),
        ("f_channels", C.c_int64),
        ("f_nhwc", C.c_int32),
        ("nhwc_pad", C.c_int32),
        ("coef", C.c_void_p),
    ]


MULTISTEP_MAX_HIST = 7


class AzMultistepArgs(C.Structure):
    _fields_ = [
        ("x_s", c_f32p),
        ("pred", c_f32p),
        ("x_t", c_f32p),
        ("mean", c_f32p),
        ("hist", c_f32p * MULTISTEP_MAX_HIST),
        ("coef", c_f32p),
        ("count", C.c_int64),
        ("n_hist", C.c_int32),
        ("pad_", C.c_int32),
    ]


class AzLinearGroup(C.Structure):
    _fields_ = [
        ("y", c_f32p),
        ("x", c_f32p),
        ("W", c_f32p),
        ("bias", c_f32p),
        ("ldy", C.c_int64),
        ("ldx", C.c_int64),
        ("N", C.c_int32),
        ("K", C.c_int32),
    ]


class AzNormFinalizeArgs(C.Structure):
    _fields_ = [
        ("S", c_f32p),
        ("T", c_f32p),
        ("partials", c_f32p),
        ("weight", c_f32p),
        ("bias", c_f32p),
        ("scale", c_f32p),
        ("shift", c_f32p),
        ("scale_bstride", C.c_int64),
        ("B", C.c_int64),
        ("C", C.c_int64),
        ("cs", C.c_int64),
        ("groups", C.c_int32),
        ("nchunks", C.c_int32),
        ("eps", C.c_float),
        ("partials1", c_f32p),
        ("quads_per_group", C.c_int32),
        ("quads0", C.c_int32),
        ("nchunks1", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class AzConvArgs(C.Structure):
    _fields_ = [
        ("src0", c_f32p),
        ("src1", c_f32p),
        ("c0s", C.c_int32),
        ("c1s", C.c_int32),
        ("up0", C.c_int32),
        ("up1", C.c_int32),
        ("h0", C.c_int32),
        ("w0", C.c_int32),
        ("h1", C.c_int32),
        ("w1", C.c_int32),
        ("batch", C.c_int32),
        ("hin", C.c_int32),
        ("win", C.c_int32),
        ("weight", c_f32p),
        ("bias", c_f32p),
        ("cout_s", C.c_int32),
        ("ksize", C.c_int32),
        ("stride", C.c_int32),
        ("pad", C.c_int32),
        ("hout", C.c_int32),
        ("wout", C.c_int32),
        ("act", C.c_int32),
        ("gate", c_f32p),
        ("gate_bstride", C.c_int64),
        ("res", c_f32p),
        ("res_up", C.c_int32),
        ("hres", C.c_int32),
        ("wres", C.c_int32),
        ("res_bcast", C.c_int32),
        ("dst", c_f32p),
        ("dst_nchw", C.c_int32),
        ("dst_c", C.c_int32),
        ("splitk", C.c_int32),
        ("workspace", c_f32p),
        ("pad_mode", C.c_int32),
        ("gn_chunks", C.c_int32),
        ("gn_quads", c_f32p),
        ("aniso", C.c_int32),
        ("stride_w", C.c_int32),
        ("up0_w", C.c_int32),
        ("up1_w", C.c_int32),
        ("in_affine", c_f32p),
        ("in_act", C.c_int32),
        ("src_dtype", C.c_int32),
        ("depth", C.c_int32),
        ("depth_shift", C.c_int32),
        ("qk_head_dim", C.c_int32),
        ("qk_heads", C.c_int32),
        ("qk_tokens", C.c_int32),
        ("qk_rmsnorm", C.c_int32),
        ("qk_eps", C.c_float),
        ("qk_reserved", C.c_int32),
        ("qk_q_weight", c_f32p),
        ("qk_k_weight", c_f32p),
        ("qk_rope_cos", c_f32p),
        ("qk_rope_sin", c_f32p),
        ("depth_wrap", C.c_int32),
        ("dst_dtype", C.c_int32),
        ("w_scale", C.c_float),
        ("reserved1", C.c_int32),
        ("in_absmax0", c_f32p),
        ("in_absmax1", c_f32p),
    ]


class AzAttnArgs(C.Structure):
    _fields_ = [
        ("q", c_f32p),
        ("k", c_f32p),
        ("v", c_f32p),
        ("out", c_f32p),
        ("batch", C.c_int32),
        ("heads", C.c_int32),
        ("tokens", C.c_int32),
        ("head_dim", C.c_int32),
        ("q_bstride", C.c_int64),
        ("q_tstride", C.c_int64),
        ("q_hstride", C.c_int64),
        ("k_bstride", C.c_int64),
        ("k_tstride", C.c_int64),
        ("k_hstride", C.c_int64),
        ("v_bstride", C.c_int64),
        ("v_tstride", C.c_int64),
        ("v_hstride", C.c_int64),
        ("o_bstride", C.c_int64),
        ("o_tstride", C.c_int64),
        ("o_hstride", C.c_int64),
        ("scale", C.c_float),
        ("qk_rmsnorm", C.c_int32),
        ("eps", C.c_float),
        ("norm_dim", C.c_int32),
        ("rope_cos", c_f32p),
        ("rope_sin", c_f32p),
        ("q_weight", c_f32p),
        ("k_weight", c_f32p),
        ("mask", c_f32p),
        ("mask_bstride", C.c_int64),
        ("mask_hstride", C.c_int64),
        ("io_dtype", C.c_int32),
        ("reserved", C.c_int32),
    ]


i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

# name -> argtypes.  Every symbol listed here MUST be exported by the shared object
# (tests/test_cabi.py checks the list against include/azula_amd.h).
PROTOTYPES: dict[str, list] = {
    "az_version": [],
    "az_step_begin": [vp, vp, vp, i32, c_stream],
    "az_transition_f32": [C.POINTER(AzTransitionArgs), c_stream],
    "az_multistep_f32": [C.POINTER(AzMultistepArgs), c_stream],
    "az_scale_f32": [vp, vp, vp, i64, c_stream],
    "az_silu_f32": [vp, vp, i64, c_stream],
    "az_axpby_f32": [vp, vp, vp, vp, vp, i64, i64, i32, c_stream],
    "az_cfg_combine_f32": [vp, vp, vp, vp, i64, c_stream],
    "az_nchw_to_nhwc_f32": [vp, vp, vp, i64, i64, i64, i64, c_stream],
    "az_nhwc_to_nchw_f32": [vp, vp, i64, i64, i64, i64, c_stream],
    "az_upsample_nearest_f32": [vp, vp, i64, i64, i64, i64, i32, i32, i64, i64, c_stream],
    "az_linear_small_f32": [vp, i64, vp, i64, vp, vp, i64, i64, i64, i32, i32, c_stream],
    "az_gather_rows_f32": [vp, vp, vp, i64, i64, i64, c_stream],
    "az_gather_step_row_f32": [vp, vp, vp, i32, i64, i64, c_stream],
    "az_coef_c_time_f32": [vp, vp, c_stream],
    "az_groupnorm_stats_f32": [vp, vp, vp, i64, i64, i64, i64, i64, i32, i32, c_stream],
    "az_groupnorm_finalize_f32": [C.POINTER(AzNormFinalizeArgs), c_stream],
    "az_affine_act_f32": [vp, vp, vp, i64, vp, vp, i64, i64, i64, i64, i32, i32, c_stream],
    "az_rownorm_mod_f32": [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, f32, c_stream],
    "az_rownorm_mod_h16": [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, f32, i32, c_stream],
    "az_groupnorm_stats_h16": [vp, vp, vp, i64, i64, i64, i64, i64, i32, i32, i32, c_stream],
    "az_affine_act_h16": [vp, vp, vp, i64, vp, vp, i64, i64, i64, i64, i32, i32, i32, c_stream],
    "az_token_copy_f32": [vp, i64, i64, vp, i64, i64, i64, i64, i64, c_stream],
    "az_token_fill_f32": [vp, i64, i64, i64, vp, i64, vp, i64, i64, c_stream],
    "az_token_fill_h16": [vp, i64, i64, i64, vp, i64, vp, i64, i64, i32, c_stream],
    "az_timestep_embedding_f32": [vp, i64, vp, i64, i64, i32, f32, c_stream],
    "az_linear_small_grouped_f32": [vp, i32, i32, i64, i32, i32, c_stream],
    "az_conv2d_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_conv2d_stem_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_conv2d_bf16_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_conv2d_x3_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_conv2d_f16_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_pack_conv_weight_half_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, c_stream],
    "az_pack_conv_weight_x3_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, c_stream],
    "az_conv2d_suggest_splitk": [i64, i32, i32, i32],
    "az_conv2d_x3_suggest_splitk": [C.POINTER(AzConvArgs)],
    "az_conv2d_winograd_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_conv2d_winograd_suggest_splitk": [i64, i32, i32, i32, i32],
    "az_winograd_pack_filter_f32": [vp, vp, i32, i32, i32, i32, i32, i32, c_stream],
    "az_conv2d_winograd_x3_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_winograd_pack_filter_x3_f32": [vp, vp, i32, i32, i32, i32, i32, i32, c_stream],
    "az_conv2d_f16x2_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_conv2d_winograd_f16x2_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_pack_conv_weight_f16x2_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, c_stream],
    "az_winograd_pack_filter_f16x2_f32": [vp, vp, i32, i32, i32, i32, i32, i32, f32, c_stream],
    "az_f16x2_weight_scale": [f32, i32],
    "az_absmax_f32": [vp, vp, i64, c_stream],
    "az_absmax_from_moments_f32": [vp, vp, i64, c_stream],
    "az_conv2d_winograd4_f32": [C.POINTER(AzConvArgs), c_stream],
    "az_conv2d_winograd4_suggest_splitk": [i64, i32, i32, i32, i32],
    "az_winograd4_pack_filter_f32": [vp, vp, i32, i32, i32, i32, i32, i32, c_stream],
    "az_pack_conv_weight_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, c_stream],
    "az_attention_f32": [C.POINTER(AzAttnArgs), c_stream],
    "az_attention_x3_f32": [C.POINTER(AzAttnArgs), c_stream],
    "az_attention_f16x2_f32": [C.POINTER(AzAttnArgs), c_stream],
    "az_attention_bf16_f32": [C.POINTER(AzAttnArgs), c_stream],
    "az_attention_f16_f32": [C.POINTER(AzAttnArgs), c_stream],
    "az_swiglu_f32": [vp, vp, i64, i64, i64, i64, c_stream],
    "az_patchify_f32": [vp, vp, vp, i64, i64, i64, i64, i64, i64, c_stream],
    "az_unpatchify_f32": [vp, vp, i64, i64, i64, i64, i64, i64, c_stream],
    "az_graph_begin": [c_stream],
    "az_graph_end": [c_stream, C.POINTER(vp)],
    "az_graph_launch": [vp, c_stream],
    "az_graph_destroy": [vp],
    "az_graph_num_nodes": [vp, C.POINTER(i64)],
    "az_transition_f64": [C.POINTER(AzTransitionArgs), c_stream],
    "az_step_row_f64": [vp, vp, vp, i32, i32, c_stream],
    "az_axpby_f64": [vp, vp, vp, vp, vp, i32, i64, i64, i32, c_stream],
    "az_randn_slice_f32": [vp, C.c_uint64, C.c_uint64, i64, i64, i64, c_stream],
    "az_scale_f64_to_f32": [vp, vp, vp, i64, i64, i32, c_stream],
    "az_calib_read_f32": [vp, vp, i64, i32, i32, i64, c_stream],
    "az_calib_write_f32": [vp, i64, f32, c_stream],
    "az_calib_mfma_f32": [vp, i32, i32, f32, f32, c_stream],
    "az_calib_mfma_random_f32": [vp, i32, i32, f32, f32, c_stream],
    "az_calib_mfma_random_bf16": [vp, i32, i32, f32, f32, c_stream],
}

RESTYPES = {"az_f16x2_weight_scale": C.c_float}  # (everything else returns an int status)

_lock = threading.Lock()
_lib = None


class AzulaAmdError(RuntimeError):
    pass


def lib() -> C.CDLL:
    r"""Loads the shared object (once).  Raises loudly if it is absent: there is no CPU path."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise AzulaAmdError(
                        f"{LIB_PATH} is missing: build it with `python -m azula_amd.csrc.build` "
                        "(hipcc, gfx950).  azula_amd has no CPU/eager fallback for device tensors."
                    )
                handle = C.CDLL(LIB_PATH)
                handle.az_error_string.restype = C.c_char_p
                handle.az_error_string.argtypes = [C.c_int]
                for name, argtypes in PROTOTYPES.items():
                    fn = getattr(handle, name)  # AttributeError if the symbol is not exported
                    fn.argtypes = argtypes
                    fn.restype = RESTYPES.get(name, C.c_int)
                _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().az_error_string(rc).decode()
        raise AzulaAmdError(f"{what} failed with code {rc}: {msg}")


def stream_ptr(device: torch.device | None = None) -> int:
    r"""Raw ``hipStream_t`` of torch's current stream (kernels are stream-ordered with torch ops)."""
    return torch.cuda.current_stream(device).cuda_stream


def on_device(fn):
    r"""Method decorator: runs ``fn(self, x, ...)`` with ``x``'s GPU as the current device.  Kernels and graphs are
    launched on ``torch.cuda.current_stream()``, i.e. on the CURRENT device's stream; with the latent on ``cuda:1``
    and ``cuda:0`` current they would otherwise run on GPU 0 against GPU-1 pointers, unordered with the torch ops on
    GPU 1's stream.  The reference works on any device index (``azula/sample.py:139-161`` is device agnostic)."""

    @functools.wraps(fn)
    def wrapped(self, x, *args, **kwargs):
        if torch.is_tensor(x) and x.is_cuda and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return fn(self, x, *args, **kwargs)
        return fn(self, x, *args, **kwargs)

    return wrapped


def ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int64, torch.int32, torch.uint8), (t.device, t.dtype)
    assert t.is_contiguous()
    return t.data_ptr()


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)
