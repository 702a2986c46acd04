r"""Guidance (reference ``azula/guidance``): classifier-free guidance on the HIP path."""

from .cfg import CFGDenoiser  # noqa: F401
