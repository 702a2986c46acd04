r"""Backbones executed by gfx950 HIP kernels (drop-ins for ``azula.nn``)."""

from . import layers, utils  # noqa: F401
from . import attention, dit, unet, vit  # noqa: F401  (the reference's module paths)
from .unet import UNet, UNetBlock  # noqa: F401
from .vit import DiT, DiTBlock, MultiheadSelfAttention, ViT  # noqa: F401
from .wrappers import TimeModulated  # noqa: F401
