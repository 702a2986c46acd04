r"""Module path of the reference's ``azula.nn.attention`` (``MultiheadSelfAttention``, reference ``attention.py:17-156``).
Inside a DiT / ViT the layer's parameters are consumed by the compiled token plan (fused-QKV MFMA GEMM + ``az_attention_f32``);
called on its own, its ``forward`` compiles a one-layer plan of the same kernels, see :mod:`azula_amd.nn.vit`."""

from .vit import MultiheadSelfAttention  # noqa: F401

__all__ = ["MultiheadSelfAttention"]
