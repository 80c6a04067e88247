"""CPU: the product library loads and exports every symbol include/guetzli_amd.h declares,
argument errors are reported as codes, and -- without a GPU -- creating a context fails
loudly with GZ_E_NO_DEVICE instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from guetzli_amd import build as gzbuild
from guetzli_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "guetzli_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gz_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    return gzbuild.build()


def test_every_declared_symbol_is_exported(lib_path):
    names = declared_functions()
    assert len(names) >= 25
    lib = C.CDLL(lib_path)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(capi.SIGNATURES) == names, "capi.SIGNATURES out of sync with the header"


def test_argument_errors_and_no_silent_fallback(lib_path):
    L = capi.Library(lib_path)
    assert L.lib.gz_abi_version() == 1
    assert L.lib.gz_strerror(-2).decode() == "no usable HIP device"
    err = C.c_int(0)
    rgb = np.zeros((16, 16, 3), np.uint8)
    assert not L.lib.gz_create(0, 4, 16, rgb.ctypes.data, 1.0, C.byref(err))
    assert err.value == -1                      # GZ_E_ARG: w < 8
    assert not L.lib.gz_create(0, 16, 16, None, 1.0, C.byref(err))
    assert err.value == -1
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        assert not L.lib.gz_create(0, 16, 16, rgb.ctypes.data, 1.0, C.byref(err))
        assert err.value == -2                  # GZ_E_NO_DEVICE, never a CPU path
        with pytest.raises(capi.GuetzliAmdError):
            L.context(rgb, 1.0)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(capi.GuetzliAmdError):
        capi.Library(str(tmp_path / "nope.so"))
