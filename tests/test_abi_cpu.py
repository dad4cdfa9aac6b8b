"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/kt_engine.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from kube_throttler_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    E.build()
    return C.CDLL(E.LIB_PATH)


def declared_symbols():
    with open(os.path.join(ROOT, "include", "kt_engine.h")) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(built_lib, s), f"{s} declared in include/kt_engine.h but not exported"
    assert set(E.EXPORTS) == set(syms)


def test_version_string(built_lib):
    built_lib.kt_version.restype = C.c_char_p
    assert b"gfx950" in built_lib.kt_version()


def test_no_cpu_fallback():
    """On a box without a GPU engine creation must fail loudly, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.EngineError) as ei:
        E.Engine(4, 4, 16, 4, 2)
    assert ei.value.code == -6


def test_invalid_config_rejected(built_lib):
    cfg = E.KtConfig(0, 4, 16, 4, 2, -1, 0)
    h = C.c_void_p()
    assert E.lib().kt_engine_create(C.byref(cfg), C.byref(h)) == -1
