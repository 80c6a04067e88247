"""The reference-side binding of INTEGRATION.md, compiled and run: the UNMODIFIED reference
`guetzli::ProcessJpegData` (built from /root/reference by tests/integration/Makefile -- test
infrastructure, like oracle/_ref) drives the C-ABI library through
`HipButteraugliComparator : guetzli::Comparator` (tests/integration/hip_comparator.{h,cc}) and
must emit the same JPEG and --verbose trace as the reference with its own comparator."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import images
from checkers import ref

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "emu"))
BUILD = os.path.join(HERE, "integration", "_build")


def _load(path):
    lib = C.CDLL(path)
    lib.gzi_process.restype = C.c_long
    lib.gzi_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_long,
                                C.c_void_p, C.c_long, C.c_void_p]
    return lib


def _process(lib, rgb, target):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    cap = 3 * w * h + (1 << 16)
    out = np.zeros(cap, np.uint8)
    tr = C.create_string_buffer(1 << 22)
    calls = (C.c_long * 2)()
    n = lib.gzi_process(rgb.ctypes.data, w, h, target, 0, out.ctypes.data, cap, tr, len(tr), calls)
    assert 0 <= n <= cap, n
    return out[:n].tobytes(), tr.value.decode(), (calls[0], calls[1])


def _make(gz_lib, out):
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "integration"), f"GZ_LIB={gz_lib}", f"OUT={out}"],
                       check=True)
    return os.path.exists(out)


@pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")
def test_reference_processor_through_the_comparator_seam_in_emulation():
    import build_emu
    out = os.path.join(BUILD, "libgz_integration_emu.so")
    if not _make(build_emu.build(), out):
        pytest.skip("tests/integration/_build not built (needs /root/reference)")
    lib = _load(out)
    rgb = images.crop(40, 32, 100, 60)
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process(rgb, target, want_trace=True)
    got_jpg, got_trace, calls = _process(lib, rgb, target)
    assert got_trace == exp_trace
    assert got_jpg == exp_jpg
    assert calls[0] >= 3 and calls[1] > 100   # Compare and CompareBlock went through the device ABI


@pytest.mark.gpu
def test_reference_processor_through_the_comparator_seam_bees():
    """guetzli::ProcessJpegData(params, jpg, &HipButteraugliComparator, ...) on tests/bees.png,
    --quality 95: the golden JPEG of BASELINE config 0 and the reference's trace."""
    import guetzli_amd
    from guetzli_amd import build as gzbuild
    out = os.path.join(BUILD, "libgz_integration.so")
    if not _make(gzbuild.LIB, out):
        pytest.skip("tests/integration/_build not built (needs /root/reference)")
    assert guetzli_amd.load().device_count() >= 1
    lib = _load(out)
    rgb = images.bees()
    jpg, trace, calls = _process(lib, rgb, 0.971769)
    assert hashlib.sha256(jpg).hexdigest() == "f2673f12a4856e020627fa151493a80b1cb2ee4dc81e28afc62dc089baf50242"
    assert hashlib.sha256(trace.encode()).hexdigest() == "954ec7623366bc3c345fc5b0748017f9a5e0128aba0917a249cca390a615f787"
    assert calls[0] > 100 and calls[1] > 250000
