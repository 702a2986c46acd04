r"""Plugins: pre-trained model zoo interface (reference ``azula/plugins``)."""

from .utils import load_cards  # noqa: F401
