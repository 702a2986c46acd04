r"""Plugin helpers: model cards (counterpart of ``azula/plugins/utils.py:29-60``)."""

from __future__ import annotations

import sys
from pathlib import Path
from types import ModuleType, SimpleNamespace

import yaml

__all__ = ["load_cards"]


def load_cards(plugin: ModuleType | str) -> dict[str, SimpleNamespace]:
    r"""Returns ``{name: card}`` for the pre-trained models of a plugin; every card has ``url``,
    ``hash`` and ``config`` (keyword arguments of the plugin's ``make_model``).

    The plugin's ``cards.yaml`` may factor hyper-parameters shared by all cards into a top-level
    ``common`` mapping; they are merged under each card's own ``config``.  A top-level ``url_format``
    (``str.format`` template) turns each card's ``url`` mapping into the full address.
    """
    module = sys.modules[plugin] if isinstance(plugin, str) else plugin
    path = Path(module.__file__).with_name("cards.yaml")
    if not path.is_file():
        raise AssertionError(f"{module} is not a plugin (no cards.yaml next to it)")
    doc = yaml.safe_load(path.read_text())
    shared = doc.get("common") or {}
    template = doc.get("url_format")
    out: dict[str, SimpleNamespace] = {}
    for name, entry in doc["cards"].items():
        url = entry["url"]
        if template is not None and not isinstance(url, str):
            url = template.format(**url)
        out[name] = SimpleNamespace(url=url, hash=entry.get("hash"), config={**shared, **entry.get("config", {})})
    return out
