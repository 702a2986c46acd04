r"""Plugin helpers (reference ``azula/plugins/utils.py:29-60``)."""

from __future__ import annotations

import os
import sys
from types import ModuleType, SimpleNamespace

import yaml

__all__ = ["load_cards"]


def load_cards(plugin: ModuleType | str) -> dict[str, SimpleNamespace]:
    r"""Name -> card (``url``, ``hash``, ``config``) mapping read from the plugin's ``cards.yaml``."""
    if isinstance(plugin, str):
        plugin = sys.modules[plugin]
    file = os.path.join(os.path.dirname(plugin.__file__), "cards.yaml")
    assert os.path.exists(file), f"{plugin} is not a plugin"
    with open(file) as f:
        cards = yaml.safe_load(f)
    return {name: SimpleNamespace(**card) for name, card in cards.items()}
