r"""pytest configuration: the ``gpu`` marker and golden-fixture helpers."""

import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (MI355X); run with -m gpu on the GPU box")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_memory():
    r"""Plans hold tens of GB of HBM at the full-size configurations; reference cycles (modules, closures) would keep a finished
    test's plans alive until the cycle collector runs, and the suite's peak would be the sum of its tests."""
    yield
    if torch.cuda.is_available():
        import gc

        gc.collect()
        torch.cuda.empty_cache()


class Golden:
    def __init__(self, name: str) -> None:
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}

    def __getitem__(self, k: str) -> torch.Tensor:
        return self.arrays[k]

    def __contains__(self, k: str) -> bool:
        return k in self.arrays


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name: str) -> Golden:
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


def max_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


# C-ABI entries by kernel family, whatever AZ_FP32_MFMA mode built the plan (native / bf16x3 / f16x2): tests that check WHICH
# kernel family a layer landed on use these
WINO_OPS = ("az_conv2d_winograd_f32", "az_conv2d_winograd_x3_f32", "az_conv2d_winograd_f16x2_f32")
DIRECT_OPS = ("az_conv2d_f32", "az_conv2d_x3_f32", "az_conv2d_f16x2_f32")
ATTN_OPS = ("az_attention_f32", "az_attention_x3_f32", "az_attention_f16x2_f32")
