r"""Module path of the reference's ``azula.nn.attention`` (``MultiheadSelfAttention``).  On the GPU the layer is
never called on its own: its parameters are consumed by the compiled DiT / ViT plans (fused-QKV MFMA GEMM +
``az_attention_f32``), see :mod:`azula_amd.nn.vit`."""

from .vit import MultiheadSelfAttention  # noqa: F401

__all__ = ["MultiheadSelfAttention"]
