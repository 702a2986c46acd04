r"""Model cache lookup (reference ``azula/hub.py``).  There is no network on the build or GPU
boxes, so this module only resolves the reference's cache layout; it never downloads."""

from __future__ import annotations

import os
import re

AZULA_HUB: str = os.path.expanduser(os.environ.get("AZULA_HUB", "~/.cache/azula/hub"))


def get_hub_dir() -> str:
    return AZULA_HUB


def set_hub_dir(cache_dir: str) -> None:
    global AZULA_HUB
    AZULA_HUB = os.path.abspath(os.path.expanduser(cache_dir))


def cached_path(url: str) -> str:
    r"""File name the reference's ``download(url)`` would use (``azula/hub.py:62-64``)."""
    return os.path.join(get_hub_dir(), re.sub(r"[^a-zA-Z0-9_]+", ".", url))


def download(url: str, hash_prefix: str | None = None, **_) -> str:
    path = cached_path(url)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found.  azula_amd does not download weights (no network): place the checkpoint from "
            f"{url} at that path (the reference's own cache location), or build a randomly initialised model "
            "with make_model(**load_cards(plugin)[name].config)."
        )
    return path
