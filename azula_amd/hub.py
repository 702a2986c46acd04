r"""Model cache (counterpart of ``azula/hub.py``) -- WITHOUT the network half.

The build and GPU boxes have no network, so :func:`download` never fetches anything: it resolves the
reference's cache layout (``~/.cache/azula/hub/<sanitised url>``, ``hub.py:62-64``), verifies the optional
``"alg:prefix"`` hash of the cached file and unpacks archives next to it exactly as the reference does
(``<file>+x``), so a cache populated by the reference -- or by hand -- is picked up unchanged.  A missing file
raises ``FileNotFoundError`` naming the path to fill.
"""

from __future__ import annotations

import hashlib
import os
import re
import shutil
import sys
import tarfile
import tempfile
import zipfile

__all__ = ["get_hub_dir", "set_hub_dir", "cached_path", "download"]

AZULA_HUB: str = os.path.expanduser(os.environ.get("AZULA_HUB", "~/.cache/azula/hub"))


def get_hub_dir() -> str:
    r"""Cache directory for models and weights."""
    return AZULA_HUB


def set_hub_dir(cache_dir: str) -> None:
    global AZULA_HUB
    AZULA_HUB = os.path.abspath(os.path.expanduser(cache_dir))


def cached_path(url: str) -> str:
    r"""File name the reference's ``download(url)`` uses: every run of characters outside ``[a-zA-Z0-9_]``
    becomes a dot (``azula/hub.py:62-64``)."""
    return os.path.join(get_hub_dir(), re.sub(r"[^a-zA-Z0-9_]+", ".", url))


def _check_hash(filename: str, hash_prefix: str) -> None:
    alg, prefix = hash_prefix.split(":")
    digest = hashlib.new(alg)
    with open(filename, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            digest.update(block)
    found = digest.hexdigest()
    if not found.startswith(prefix):
        raise AssertionError(
            f"The hash of the cached file ({alg}:{found}) does not match the expected hash prefix ({alg}:{prefix})."
        )


def download(
    url: str,
    filename: str | None = None,
    hash_prefix: str | None = None,
    extract: bool = False,
    quiet: bool = False,
) -> str:
    r"""Resolves ``url`` in the cache (same signature and return value as the reference's ``download``).

    Returns the local file name, or with ``extract=True`` the directory ``<file>+x`` holding the unpacked
    archive (created on first use from the cached tar / zip file)."""
    filename = cached_path(url) if filename is None else os.path.abspath(os.path.expanduser(filename))
    if not os.path.exists(filename):
        if extract and os.path.isdir(f"{filename}+x"):
            return f"{filename}+x"  # archive removed after unpacking: the unpacked tree is all that is needed
        raise FileNotFoundError(
            f"{filename} not found.  azula_amd does not download weights (no network): place the file from {url} "
            "at that path (the reference's own cache location), or build a randomly initialised model with "
            "make_model(**load_cards(plugin)[name].config)."
        )
    if not quiet:
        print(f"Loading from {filename}", file=sys.stderr)
    if hash_prefix is not None:
        _check_hash(filename, hash_prefix)
    if not extract:
        return filename
    xd = f"{filename}+x"
    if os.path.exists(xd):
        return xd
    with tempfile.TemporaryDirectory(dir=os.path.dirname(filename)) as td:
        if tarfile.is_tarfile(filename):
            with tarfile.open(filename, "r") as f:
                f.extractall(td)
        elif zipfile.is_zipfile(filename):
            with zipfile.ZipFile(filename, "r") as f:
                f.extractall(td)
        else:
            raise ValueError("Unknown archive format.")
        os.makedirs(xd)
        for name in os.listdir(td):
            shutil.move(os.path.join(td, name), os.path.join(xd, name))
    return xd
