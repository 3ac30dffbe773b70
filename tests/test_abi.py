"""CPU-side checks of the C-ABI boundary: liboffk.so builds, loads, and exports exactly
the symbols include/offk.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

import offk_amd  # noqa: F401
from offk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "offk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(offk_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return _lib.load()


def test_header_and_binding_agree():
    syms = header_symbols()
    assert len(syms) >= 18
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol(built):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(raw, s), s
    assert built.offk_abi_version() == 10


def test_handleless_errors_are_reported(built):
    # argument validation happens before any HIP call, so this is safe without a GPU
    rc = built.offk_conv2d(None, None, 0, 0, 0, 0, 0, 0, None, None, 0, 0, 0, 0, 0, None, 0, 0, 0, None, 0, 0)
    assert rc == -1
    assert b"offk_conv2d" in built.offk_last_error(None)
    rc = built.offk_create(None, None)
    assert rc == -1


def test_create_without_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = _lib.OffkConfig(1, 7, 0, 0, 0, 101, 0, 0, 0)
    h = ctypes.c_void_p()
    rc = built.offk_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -5 and not h.value
    from offk_amd import runtime
    with pytest.raises(_lib.OffkError):
        runtime.OffForward(1, 7)
