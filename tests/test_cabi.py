"""The C-ABI library loads and exports every symbol include/fvhd.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ml_fastvlm_amd import build, _lib
    build.build_library()
    return _lib.load()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "fvhd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fvhd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = _declared_functions()
    assert len(names) >= 20, names
    raw = ctypes.CDLL(os.path.join(ROOT, "ml_fastvlm_amd", "libfvhd.so"))
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, f"declared in fvhd.h but not exported: {missing}"


def test_binding_covers_header(lib):
    # every declared function has ctypes argtypes in the Python stub
    for n in _declared_functions():
        assert getattr(lib, n).argtypes is not None or n in ("fvhd_version", "fvhd_last_error"), n


def test_version_and_error_string(lib):
    assert lib.fvhd_version() >= 100
    assert isinstance(lib.fvhd_last_error(), bytes)


def test_bad_arguments_are_errors_not_crashes(lib):
    h = ctypes.c_void_p()
    assert lib.fvhd_create(ctypes.byref(h), 0, 1000, 1) != 0          # not a multiple of 64
    assert b"multiple of 64" in lib.fvhd_last_error()
    assert lib.fvhd_create(ctypes.byref(h), 0, 1024, 0) != 0
    assert lib.fvhd_finalize_weights(None) != 0
    assert lib.fvhd_encode(None, None, 0, 1, None, 0, None) != 0
    assert lib.fvhd_num_tokens(None) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(lib):
    h = ctypes.c_void_p()
    assert lib.fvhd_create(ctypes.byref(h), 0, 1024, 1) != 0
    assert b"no HIP device" in lib.fvhd_last_error()
    assert not h.value
