"""CPU tests of the host side (guetzli_amd/host): the JPEG writer byte-for-byte against
the reference's WriteJpeg, the quality table, and -- with the product kernels running in
the test-suite's CPU emulation -- a WHOLE encode byte-for-byte (and --verbose trace line
for line) against the unmodified reference guetzli::Process."""
import hashlib
import os
import sys

import numpy as np
import pytest

import images
from checkers import ref

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import build_emu  # noqa: E402
from guetzli_amd import build as gzbuild  # noqa: E402
from guetzli_amd.encoder import HostLibrary  # noqa: E402

needs_ref = pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")


@pytest.fixture(scope="module")
def host():
    gzbuild.build()
    return HostLibrary(gzbuild.build_host())


@pytest.fixture(scope="module")
def host_emu():
    return HostLibrary(build_emu.build_host())


def test_quality_table(host):
    assert abs(host.butteraugli_score_for_quality(95) - 0.971769) < 1e-12
    assert abs(host.butteraugli_score_for_quality(84) - 1.945456) < 1e-12
    if ref is not None:
        for q in (70, 84, 84.5, 90.25, 95, 100, 110, 150, 10):
            assert host.butteraugli_score_for_quality(q) == \
                ref._butteraugli_score_for_quality(q)


@needs_ref
@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (33, 40), (8, 8)])
def test_write_jpeg_bytes(host, wh):
    w, h = wh
    rng = np.random.default_rng(w * 1000 + h)
    rgb = images.crop(w, h)
    co = ref.encode_rgb(rgb)
    qs = [np.ones((3, 64), np.int32),
          np.full((3, 64), 3, np.int32),
          np.stack([rng.integers(1, 9, 64), rng.integers(1, 30, 64), rng.integers(1, 30, 64)]).astype(np.int32),
          np.stack([rng.integers(1, 9, 64)] * 3).astype(np.int32),
          np.stack([rng.integers(200, 400, 64), rng.integers(1, 30, 64), rng.integers(1, 30, 64)]).astype(np.int32)]
    for q in qs:
        exp = ref.write_jpeg(co, w, h, q)
        cq, _, _ = ref.reconstruct(co, w, h, q)
        got = host.write_jpeg(cq, w, h, q)
        assert got == exp, (len(got), len(exp))
    # grayscale image: chroma all zero -> single-component frame
    gray = np.repeat(rgb[:, :, 1:2], 3, axis=2)
    cg = ref.encode_rgb(gray)
    q = np.full((3, 64), 2, np.int32)
    cq, _, _ = ref.reconstruct(cg, w, h, q)
    if not cq[1:].any():
        assert host.write_jpeg(cq, w, h, q) == ref.write_jpeg(cg, w, h, q)


@needs_ref
@pytest.mark.parametrize("case", [(40, 32, 100, 60, 95, None), (48, 40, 300, 150, 84, None),
                                  (40, 32, 100, 60, 95, 128)])
def test_whole_encode_matches_reference_in_emulation(host_emu, case, monkeypatch):
    w, h, x0, y0, quality, dev_threshold = case
    if dev_threshold:   # partition even these small orders with the device kernels
        monkeypatch.setenv("GZ_ORDER_DEVICE_THRESHOLD", str(dev_threshold))
    rgb = images.crop(w, h, x0, y0)
    target = ref._butteraugli_score_for_quality(float(quality))
    exp_jpg, exp_trace = ref.process(rgb, target, want_trace=True)
    got_jpg, info = host_emu.process(rgb, quality=quality, want_trace=True)
    exp_lines, got_lines = exp_trace.splitlines(), info["trace"].splitlines()
    for i, (a, b) in enumerate(zip(exp_lines, got_lines)):
        assert a == b, f"trace line {i}:\n ref: {a}\n got: {b}"
    assert len(exp_lines) == len(got_lines)
    assert hashlib.sha256(got_jpg).hexdigest() == hashlib.sha256(exp_jpg).hexdigest()
    assert info["counters"]["number of iterations"] >= 3


@needs_ref
@pytest.mark.parametrize("wh", [(24, 40), (31, 64), (8, 8), (1, 1), (5, 3)])
def test_small_images_emit_the_unquantised_jpeg(host_emu, wh):
    """w or h < 32: no butteraugli; Process() returns the q = 1 JPEG of EncodeRGBToJpeg
    (processor.cc:832-838).  Same bytes and trace as the reference."""
    w, h = wh
    rgb = images.crop(w, h, 200, 100)
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process(rgb, target, want_trace=True)
    got_jpg, info = host_emu.process(rgb, quality=95, want_trace=True)
    assert got_jpg == exp_jpg
    assert info["trace"] == exp_trace


def _pil_jpeg(rgb, **kw):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **kw)
    return b.getvalue()


@needs_ref
@pytest.mark.parametrize("case", [
    (48, 40, dict(quality=97, subsampling=0), True),
    (40, 32, dict(quality=99, subsampling=0, progressive=True, comment=b"hello"), True),
    (40, 32, dict(quality=98, subsampling=0, optimize=True, comment=b"kept", dpi=(72, 72)), False),
    (24, 40, dict(quality=95, subsampling=0), True),      # too small for butteraugli
    (24, 40, dict(quality=95, subsampling=0, comment=b"x"), False),
])
def test_jpeg_input_matches_reference_in_emulation(host_emu, case):
    """guetzli::Process(params, stats, jpeg_data, &out) (processor.cc:890-924) for 4:4:4
    input: the same bytes and the same --verbose trace as the reference, with and without
    clear_metadata."""
    w, h, kw, clear = case
    data = _pil_jpeg(images.crop(w, h, 100, 60), **kw) + (b"" if clear else b"tail!")
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process_jpeg(data, target, clear_metadata=clear, want_trace=True)
    assert exp_jpg is not None
    got_jpg, got_trace = host_emu.process_jpeg(data, quality=95, clear_metadata=clear, want_trace=True)
    for i, (a, b) in enumerate(zip(exp_trace.splitlines(), got_trace.splitlines())):
        assert a == b, f"trace line {i}:\n ref: {a}\n got: {b}"
    assert got_trace == exp_trace
    assert got_jpg == exp_jpg


@needs_ref
def test_jpeg_input_refusals(host_emu):
    """What the reference rejects is rejected; 4:2:0 input is refused here (not implemented)."""
    rgb = images.crop(48, 40, 100, 60)
    target = ref._butteraugli_score_for_quality(95.0)
    grey = _pil_jpeg(np.ascontiguousarray(rgb[:, :, 1]), quality=95)
    assert ref.process_jpeg(grey, target)[0] is None
    with pytest.raises(RuntimeError):
        host_emu.process_jpeg(grey, quality=95)
    with pytest.raises(RuntimeError):
        host_emu.process_jpeg(b"not a jpeg", quality=95)
    with pytest.raises(RuntimeError):
        host_emu.process_jpeg(_pil_jpeg(rgb, quality=95, subsampling=2), quality=95)
