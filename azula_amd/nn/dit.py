r"""Module path of the reference's ``azula.nn.dit`` (``DiT``, ``DiTBlock``): the implementation shares
:mod:`azula_amd.nn.vit`'s compiled token path."""

from .vit import DiT, DiTBlock  # noqa: F401

__all__ = ["DiT", "DiTBlock"]
