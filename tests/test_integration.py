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


# ---- the whole-Process seam: the UNMODIFIED front end guetzli/guetzli.cc over this repository ----
def _make_cli(host_lib, out):
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "integration"), "cli",
                        f"HOST_LIB={host_lib}", f"CLI={out}"], check=True)
    return os.path.exists(out)


def _png(rgb):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "PNG")
    return b.getvalue()


def _cli(exe, args, data, tmp_path, suffix=".png"):
    """`guetzli [flags] in out` -> (return code, output bytes or None, stderr text)."""
    src, dst = tmp_path / ("in" + suffix), tmp_path / "out.jpg"
    src.write_bytes(data)
    if dst.exists():
        dst.unlink()
    r = subprocess.run([exe] + list(args) + [str(src), str(dst)], capture_output=True, timeout=1200)
    return r.returncode, (dst.read_bytes() if dst.exists() else None), r.stderr.decode()


@pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")
def test_unmodified_front_end_over_the_drop_in_process_in_emulation(tmp_path):
    """guetzli/guetzli.cc compiled UNCHANGED, guetzli::Process (both overloads, processor.h:39-41,
    54-56) supplied by tests/integration/process_adapter.cc -> guetzli_amd::Process: PNG and JPEG
    input, --quality, --verbose (the reference's trace on stderr), the front end's own refusals."""
    import build_emu
    exe = os.path.join(BUILD, "guetzli_emu")
    if not _make_cli(build_emu.build_host(), exe):
        pytest.skip("tests/integration/_build not built (needs /root/reference)")
    rgb = images.crop(40, 32, 100, 60)
    for q in (95, 84):
        exp_jpg, exp_trace = ref.process(rgb, ref._butteraugli_score_for_quality(float(q)), want_trace=True)
        rc, got, err = _cli(exe, ["--verbose", "--quality", str(q)], _png(rgb), tmp_path)
        assert rc == 0, err
        assert got == exp_jpg
        assert err == exp_trace
    # JPEG input goes through the other overload
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(images.crop(48, 40, 30, 20)).save(b, "JPEG", quality=97, subsampling=0)
    exp_jpg, _ = ref.process_jpeg(b.getvalue(), ref._butteraugli_score_for_quality(95.0))
    rc, got, err = _cli(exe, [], b.getvalue(), tmp_path, ".jpg")
    assert rc == 0 and got == exp_jpg, err
    # the front end's own checks still guard the call (guetzli.cc:286-292)
    rc, got, err = _cli(exe, ["--memlimit", "1"], _png(rgb), tmp_path)
    assert rc == 1 and got is None and "Memory limit would be exceeded" in err
    rc, got, err = _cli(exe, [], b"\x89PNG\r\n\x1a\nnot a png", tmp_path)
    assert rc == 1 and got is None and "Error reading PNG data" in err


@pytest.mark.gpu
def test_unmodified_front_end_over_the_drop_in_process_bees(tmp_path):
    """`guetzli_hip tests/bees.png out.jpg` (BASELINE configs[0] as the reference's golden test
    runs it, tests/golden_test.sh): the golden JPEG; --quality 84; --verbose = the reference's
    trace, byte for byte, on stderr."""
    from guetzli_amd import build as gzbuild
    exe = os.path.join(BUILD, "guetzli_hip")
    if not _make_cli(gzbuild.HOST_LIB, exe):
        pytest.skip("tests/integration/_build not built (needs /root/reference)")
    data = open(images.BEES, "rb").read()
    rc, jpg, err = _cli(exe, [], data, tmp_path)
    assert rc == 0, err
    assert hashlib.sha256(jpg).hexdigest() == "f2673f12a4856e020627fa151493a80b1cb2ee4dc81e28afc62dc089baf50242"
    rc, jpg, err = _cli(exe, ["--quality", "84"], data, tmp_path)
    assert rc == 0 and hashlib.sha256(jpg).hexdigest() == "95f509f457ce8ddd85087c804664539e0ef7b1f3a6c6ca1cbedcd9ba29a89379"
    rc, jpg, err = _cli(exe, ["--verbose"], data, tmp_path)
    assert rc == 0 and hashlib.sha256(jpg).hexdigest() == "f2673f12a4856e020627fa151493a80b1cb2ee4dc81e28afc62dc089baf50242"
    assert hashlib.sha256(err.encode()).hexdigest() == "954ec7623366bc3c345fc5b0748017f9a5e0128aba0917a249cca390a615f787"
