r"""The C-ABI shared library loads and exports every symbol ``include/azula_amd.h`` declares
(no compute calls: this runs without a GPU)."""

import ctypes
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "azula_amd.h")


def declared_symbols() -> list[str]:
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from azula_amd.csrc import build

    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    handle = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(handle, s)]
    assert not missing, f"declared in azula_amd.h but not exported: {missing}"


def test_python_prototypes_match_header(built_lib):
    from azula_amd import _lib

    syms = set(declared_symbols())
    assert set(_lib.PROTOTYPES) <= syms
    assert syms - set(_lib.PROTOTYPES) <= {"az_error_string"}
    assert _lib.lib().az_version() == 1
    assert b"NULL" in _lib.lib().az_error_string(-1)


def test_struct_layouts_match_c():
    from azula_amd import _lib

    names = ["AzStepCoef", "AzTransitionArgs", "AzNormFinalizeArgs", "AzConvArgs", "AzAttnArgs"]
    prog = '#include <stdio.h>\n#include "azula_amd.h"\nint main(void){' + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names
    ) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    for n, sz in zip(names, sizes):
        assert ctypes.sizeof(getattr(_lib, n)) == sz, n


def test_missing_library_is_loud(monkeypatch):
    from azula_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libazula_amd.so")
    with pytest.raises(_lib.AzulaAmdError, match="no CPU/eager fallback"):
        _lib.lib()
