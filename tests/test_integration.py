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
    lib.gzi_process_params.restype = C.c_long
    lib.gzi_process_params.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
    return lib


def _process(lib, rgb, target, force_420=False, try_420=False):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    cap = 3 * w * h + (1 << 16)
    out = np.zeros(cap, np.uint8)
    tr = C.create_string_buffer(1 << 22)
    calls = (C.c_long * 3)()
    n = lib.gzi_process_params(rgb.ctypes.data, w, h, target, 0, int(force_420), int(try_420),
                               out.ctypes.data, cap, tr, len(tr), calls)
    assert 0 <= n <= cap, n
    return out[:n].tobytes(), tr.value.decode(), (calls[0], calls[1], calls[2])


def _make(gz_lib, out, batched=False):
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "integration"), f"GZ_LIB={gz_lib}", f"OUT={out}"] +
                       (["BATCHED=1"] if batched else []), check=True)
    return os.path.exists(out)


def _golden(name):
    import json
    return json.load(open(os.path.join(HERE, "golden", "params_hashes.json")))[name]


@pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")
@pytest.mark.parametrize("kw", [dict(), dict(force_420=True)])
def test_reference_processor_through_the_comparator_seam_in_emulation(kw):
    """4:4:4, and -- SwitchBlock / CompareBlock with factors 2 x 2, Compare on a 4:2:0 frame,
    block weights by factor -- Params::force_420 (comparator.h:50-52, processor.cc:873-877)."""
    import build_emu
    out = os.path.join(BUILD, "libgz_integration_emu.so")
    if not _make(build_emu.build(), out):
        pytest.skip("tests/integration/_build not built (needs /root/reference)")
    lib = _load(out)
    rgb = images.crop(40, 32, 100, 60)
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process_params(rgb, target, want_trace=True, **kw)
    got_jpg, got_trace, calls = _process(lib, rgb, target, **kw)
    assert got_trace == exp_trace
    assert got_jpg == exp_jpg
    assert calls[0] >= 3 and calls[1] > 100 and calls[2] == 0   # Compare and CompareBlock went through the device ABI
    if kw:
        assert "f112222" in exp_trace


@pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")
@pytest.mark.parametrize("kw", [dict(), dict(force_420=True)])
def test_patched_reference_processor_with_the_batched_hook_in_emulation(kw):
    """INTEGRATION.md section 2 compiled: the reference's processor.cc / comparator.h patched at
    build time (tests/integration/patch_reference.py) so that SelectFrequencyMasking asks the
    comparator for phase A of all blocks at once; no CompareBlock round trip is left, bytes and
    trace are the reference's."""
    import build_emu
    out = os.path.join(BUILD, "libgz_integration_batched_emu.so")
    if not _make(build_emu.build(), out, batched=True):
        pytest.skip("tests/integration/_build not built (needs /root/reference)")
    lib = _load(out)
    rgb = images.crop(40, 32, 100, 60)
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process_params(rgb, target, want_trace=True, **kw)
    got_jpg, got_trace, calls = _process(lib, rgb, target, **kw)
    assert got_trace == exp_trace
    assert got_jpg == exp_jpg
    assert calls[0] >= 3 and calls[1] == 0 and calls[2] == (2 if kw else 1)


def _gpu_lib(batched=False):
    import guetzli_amd
    from guetzli_amd import build as gzbuild
    out = os.path.join(BUILD, "libgz_integration_batched.so" if batched else "libgz_integration.so")
    if not _make(gzbuild.LIB, out, batched=batched):
        pytest.skip("tests/integration/_build not built (needs /root/reference)")
    assert guetzli_amd.load().device_count() >= 1
    return _load(out)


@pytest.mark.gpu
def test_reference_processor_through_the_comparator_seam_bees():
    """guetzli::ProcessJpegData(params, jpg, &HipButteraugliComparator, ...) on tests/bees.png,
    --quality 95: the golden JPEG of BASELINE config 0 and the reference's trace."""
    lib = _gpu_lib()
    rgb = images.bees()
    jpg, trace, calls = _process(lib, rgb, 0.971769)
    assert hashlib.sha256(jpg).hexdigest() == "f2673f12a4856e020627fa151493a80b1cb2ee4dc81e28afc62dc089baf50242"
    assert hashlib.sha256(trace.encode()).hexdigest() == "954ec7623366bc3c345fc5b0748017f9a5e0128aba0917a249cca390a615f787"
    assert calls[0] > 100 and calls[1] > 250000


@pytest.mark.gpu
def test_reference_processor_through_the_comparator_seam_bees_force_420():
    """The same with Params::force_420: the UNMODIFIED reference Processor drives the 4:2:0 round
    (Compare on the 4:2:0 frame, SwitchBlock / CompareBlock with factors 2 x 2 for the chroma
    search, block weights on the 16 x 16 grid) through the binding."""
    lib = _gpu_lib()
    g = _golden("bees_force420_q95")
    jpg, trace, calls = _process(lib, images.bees(), 0.971769, force_420=True)
    assert hashlib.sha256(jpg).hexdigest() == g["jpeg_sha256"]
    assert "f112222" in trace and calls[1] > 100000


@pytest.mark.gpu
@pytest.mark.parametrize("kw,golden", [(dict(), None), (dict(force_420=True), "bees_force420_q95")])
def test_patched_reference_processor_with_the_batched_hook_bees(kw, golden):
    """The patched reference Processor (batched phase-A hook) on tests/bees.png: the golden JPEG,
    no per-block round trip, and seconds instead of the 12 s of the per-block seam."""
    import time
    lib = _gpu_lib(batched=True)
    rgb = images.bees()
    _process(lib, rgb, 0.971769, **kw)            # warm-up (code-object load, pools)
    t0 = time.perf_counter()
    jpg, trace, calls = _process(lib, rgb, 0.971769, **kw)
    dt = time.perf_counter() - t0
    want = _golden(golden)["jpeg_sha256"] if golden else "f2673f12a4856e020627fa151493a80b1cb2ee4dc81e28afc62dc089baf50242"
    assert hashlib.sha256(jpg).hexdigest() == want
    if not golden:
        assert hashlib.sha256(trace.encode()).hexdigest() == "954ec7623366bc3c345fc5b0748017f9a5e0128aba0917a249cca390a615f787"
    assert calls[1] == 0 and calls[2] >= 1
    print(f"patched reference Processor + batched hook, bees {kw}: {dt:.2f} s")
    assert dt < 3.0, dt
