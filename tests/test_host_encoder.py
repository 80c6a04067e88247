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
@pytest.mark.parametrize("threshold", [None, 128])
def test_repeated_content_ties_in_the_global_order(host_emu, monkeypatch, threshold):
    """An image of one 16x16 patch repeated: every block has eleven exact copies, so the keys of
    phase B's global order tie in groups of twelve and which copy std::sort serves first decides the
    bytes -- the emulation's other images have no ties to get wrong (a round-4 experiment that
    rearranged small orders on the device without the host replaying it passed every CPU test and
    failed on the GPU's tiled images).  Default threshold: the host sorts these small orders itself;
    128: the device's partitions and descents, replayed by the host."""
    if threshold:
        monkeypatch.setenv("GZ_ORDER_DEVICE_THRESHOLD", str(threshold))
    rgb = np.ascontiguousarray(np.tile(images.crop(16, 16, 200, 100), (3, 4, 1)))
    assert rgb.shape == (48, 64, 3)
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process(rgb, target, want_trace=True)
    got_jpg, info = host_emu.process(rgb, quality=95, want_trace=True)
    assert info["trace"].splitlines() == exp_trace.splitlines()
    assert got_jpg == exp_jpg
    got2, info2 = host_emu.process(rgb, quality=95)      # (without a trace: the size-bound path)
    assert got2 == exp_jpg
    if threshold:
        assert info2["counters"]["phase B partitions made ahead"] > 0


def test_descent_position_mismatch_aborts_the_encode(host_emu, monkeypatch, capfd):
    """The device rearranges the order along ITS derivation of the descent position; the host
    replays the logged partitions along its own.  If the two ever disagreed the array would no
    longer be what LazySorted believes -- the driver must fail, not emit other bytes.
    GZ_EMU_SKEW_DESCENT (emulation build only) makes gz_order_descend_end report another position."""
    monkeypatch.setenv("GZ_ORDER_DEVICE_THRESHOLD", "128")
    rgb = images.crop(40, 32, 100, 60)
    _, info = host_emu.process(rgb, quality=95)
    assert info["counters"]["phase B partitions made ahead"] > 0      # the guard is on this path
    monkeypatch.setenv("GZ_EMU_SKEW_DESCENT", "1")
    with pytest.raises(RuntimeError):
        host_emu.process(rgb, quality=95)
    assert "gz_order_descend: position" in capfd.readouterr().err


@needs_ref
@pytest.mark.parametrize("case", [
    ("flat", (0, 0, 0), 40, 32, 95, {}), ("flat", (255, 255, 255), 33, 35, 95, {"try_420": True}),
    ("flat", (255, 0, 0), 48, 32, 84, {"force_420": True}), ("stripes", 5, 48, 40, 95, {}),
    ("stripes", 7, 41, 37, 90, {"force_420": True}), ("noise", 1, 40, 32, 95, {}),
    ("noise", 2, 34, 33, 84, {"try_420": True}),
])
def test_degenerate_content_matches_reference_in_emulation(host_emu, case):
    """Content at the edges of the search: flat images (all AC zero, empty zeroing
    orders, the v < 1e-4 branch of CalculateDiffmap, butteraugli.cc:722-732), saturated stripes,
    uniform noise (every coefficient a candidate) -- bytes and --verbose trace of the reference."""
    kind, arg, w, h, quality, params = case
    rgb = {"flat": lambda: images.flat(w, h, arg), "stripes": lambda: images.stripes(w, h, period=arg),
           "noise": lambda: images.noise(w, h, seed=arg)}[kind]()
    target = ref._butteraugli_score_for_quality(float(quality))
    exp_jpg, exp_trace = ref.process_params(rgb, target, want_trace=True, **params)
    got_jpg, info = host_emu.process(rgb, quality=quality, want_trace=True, **params)
    assert info["trace"].splitlines() == exp_trace.splitlines()
    assert got_jpg == exp_jpg


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
    """What the reference rejects is rejected."""
    rgb = images.crop(48, 40, 100, 60)
    target = ref._butteraugli_score_for_quality(95.0)
    grey = _pil_jpeg(np.ascontiguousarray(rgb[:, :, 1]), quality=95)
    assert ref.process_jpeg(grey, target)[0] is None
    with pytest.raises(RuntimeError):
        host_emu.process_jpeg(grey, quality=95)
    with pytest.raises(RuntimeError):
        host_emu.process_jpeg(b"not a jpeg", quality=95)
    y422 = _pil_jpeg(rgb, quality=95, subsampling=1)   # 4:2:2: neither Is444 nor Is420
    assert ref.process_jpeg(y422, target)[0] is None
    with pytest.raises(RuntimeError):
        host_emu.process_jpeg(y422, quality=95)


# ------------------------------------------------- non-default guetzli::Params (row f4) --
@needs_ref
@pytest.mark.parametrize("case", [
    (40, 32, 100, 60, 95, dict(force_420=True)),
    (48, 40, 300, 150, 84, dict(try_420=True)),
    (33, 35, 200, 100, 90, dict(force_420=True)),          # odd size: MCU padding on both axes
    (40, 32, 100, 60, 95, dict(lookahead=1)),
    (40, 32, 100, 60, 90, dict(lookahead=5, new_model=False)),
    (48, 33, 10, 10, 95, dict(try_420=True, lookahead=2)),
    (40, 32, 100, 60, 95, dict(force_420=True, use_silver_screen=True)),
    (35, 33, 200, 100, 90, dict(try_420=True, use_silver_screen=True)),   # odd size
])
def test_whole_encode_with_params_matches_reference_in_emulation(host_emu, case, monkeypatch):
    """Params::try_420 / force_420 (OutputImage::Downsample, the 4:2:0 pixel model, the
    comp_mask 1 / 6 searches, processor.cc:847-878), zeroing_greedy_lookahead and
    new_zeroing_model: bytes and --verbose trace of the unmodified reference.  The device
    entropy coder is cross-checked against the serial host writer on every candidate."""
    w, h, x0, y0, quality, kw = case
    monkeypatch.setenv("GZ_VERIFY_ENTROPY", "1")
    rgb = images.crop(w, h, x0, y0)
    target = ref._butteraugli_score_for_quality(float(quality))
    ref_kw = dict(kw)
    if "use_silver_screen" in ref_kw:
        ref_kw["silver"] = ref_kw.pop("use_silver_screen")
    exp_jpg, exp_trace = ref.process_params(rgb, target, want_trace=True, **ref_kw)
    got_jpg, info = host_emu.process(rgb, quality=quality, want_trace=True, **kw)
    for i, (a, b) in enumerate(zip(exp_trace.splitlines(), info["trace"].splitlines())):
        assert a == b, f"trace line {i}:\n ref: {a}\n got: {b}"
    assert info["trace"] == exp_trace
    assert got_jpg == exp_jpg
    if kw.get("force_420") or kw.get("try_420"):
        assert "f112222" in exp_trace and "YUV420 selected" in exp_trace


@needs_ref
@pytest.mark.parametrize("case", [(40, 32, 95, dict()), (64, 48, 90, dict()), (40, 32, 95, dict(force_420=True)),
                                  (48, 40, 90, dict(try_420=True))])
def test_grey_image_matches_reference_in_emulation(host_emu, case):
    """r = g = b input: SaveToJpegData writes one component, BuildACHistograms leaves the chroma
    statistics of phase B's size model empty (processor.cc:592-600), Downsample does nothing
    (output_image.cc:305-308) and the chroma search is skipped (:546-547)."""
    w, h, quality, kw = case
    rgb = np.repeat(images.crop(w, h, 120, 80)[:, :, 1:2], 3, axis=2).copy()
    target = ref._butteraugli_score_for_quality(float(quality))
    exp_jpg, exp_trace = ref.process_params(rgb, target, want_trace=True, **kw)
    got_jpg, info = host_emu.process(rgb, quality=quality, want_trace=True, **kw)
    for i, (a, b) in enumerate(zip(exp_trace.splitlines(), info["trace"].splitlines())):
        assert a == b, f"trace line {i}:\n ref: {a}\n got: {b}"
    assert info["trace"] == exp_trace
    assert got_jpg == exp_jpg


@needs_ref
@pytest.mark.parametrize("case", [
    (48, 40, dict(quality=97, subsampling=2), True, dict()),
    (41, 35, dict(quality=98, subsampling=2, progressive=True, comment=b"hi"), False, dict()),
    (24, 40, dict(quality=95, subsampling=2), True, dict()),          # too small for butteraugli
    (40, 32, dict(quality=97, subsampling=0), True, dict(try_420=True)),   # 4:4:4 input, both modes
    # r = g = b content in a 3-component 4:2:0 file: chroma all zero -> SaveToJpegData writes one
    # component although the frame stays 4:2:0 (ymul 1.0, no chroma search, one AC histogram)
    (48, 40, dict(quality=97, subsampling=2, grey=True), True, dict()),
    (41, 35, dict(quality=96, subsampling=2, grey=True), False, dict()),
])
def test_jpeg_420_input_matches_reference_in_emulation(host_emu, case, monkeypatch):
    """YUV 4:2:0 JPEG input (processor.cc:811-815,847-849): decoded with the 2x2 pixel model,
    the original written from the input's own (padded) blocks, then the 4:2:0 search."""
    w, h, kw, clear, params = case
    monkeypatch.setenv("GZ_VERIFY_ENTROPY", "1")
    kw = dict(kw)
    rgb = images.crop(w, h, 100, 60)
    if kw.pop("grey", False):
        rgb = np.repeat(rgb[:, :, 1:2], 3, axis=2).copy()
    data = _pil_jpeg(rgb, **kw) + (b"" if clear else b"tail!")
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process_params(data, target, clear_metadata=clear, want_trace=True, **params)
    assert exp_jpg is not None
    got_jpg, got_trace = host_emu.process_jpeg(data, quality=95, clear_metadata=clear, want_trace=True, **params)
    for i, (a, b) in enumerate(zip(exp_trace.splitlines(), got_trace.splitlines())):
        assert a == b, f"trace line {i}:\n ref: {a}\n got: {b}"
    assert got_trace == exp_trace
    assert got_jpg == exp_jpg


def test_c_wrappers_do_not_let_exceptions_through(host_emu):
    """A PNG header that declares an absurd size is rejected without allocating for it, and a
    C++ exception inside any gzh_* entry point comes back as an error code."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1000000, 1000000, 16, 6, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    with pytest.raises(ValueError):
        host_emu.read_png(bomb)


# ---- round 5: photographs and qualities off the table's tested rows, in emulation -------------------
PHOTO_CROPS = [
    # photo, x0, y0, w, h, quality, Params
    ("astronaut", 200, 60, 48, 40, 95.0, {}),            # skin, hair, a hard collar edge
    ("china", 300, 40, 40, 32, 100.0, {}),               # sky gradient against a roof line; the table's last row
    ("coffee", 250, 150, 40, 32, 97.5, {}),              # porcelain highlight; interpolated quality
    ("gravel", 100, 100, 40, 32, 85.5, {}),              # grey texture (all-zero chroma); interpolated quality
    ("chelsea", 180, 90, 48, 40, 90.0, dict(try_420=True)),   # fur
    ("hubble", 400, 300, 40, 32, 99.0, {}),              # points of light on black
    ("flower", 280, 150, 40, 32, 110.0, dict(force_420=True)),   # quality clamped to the table's end (quality.cc:79-80)
]


@needs_ref
@pytest.mark.parametrize("case", PHOTO_CROPS, ids=[f"{c[0]}_q{c[5]:g}" for c in PHOTO_CROPS])
def test_photograph_crops_match_reference_in_emulation(host_emu, case):
    """Crops of the committed photographs (tests/golden/photos) through the whole driver with the
    kernels in emulation: bytes and --verbose trace of the unmodified reference, at q100 (tiny
    target: the search ends in 'up' iterations, processor.cc:690-698), fractional qualities (the
    interpolation of quality.cc:78-85) and q110 (clamped)."""
    name, x0, y0, w, h, quality, params = case
    rgb = np.ascontiguousarray(images.photo(name)[y0:y0 + h, x0:x0 + w])
    assert rgb.shape == (h, w, 3)
    target = ref._butteraugli_score_for_quality(quality)
    assert host_emu.butteraugli_score_for_quality(quality) == target
    exp_jpg, exp_trace = ref.process_params(rgb, target, want_trace=True, **params)
    got_jpg, info = host_emu.process(rgb, quality=quality, want_trace=True, **params)
    exp_lines, got_lines = exp_trace.splitlines(), info["trace"].splitlines()
    for i, (a, b) in enumerate(zip(exp_lines, got_lines)):
        assert a == b, f"trace line {i}:\n ref: {a}\n got: {b}"
    assert len(exp_lines) == len(got_lines)
    assert got_jpg == exp_jpg
    got2, _ = host_emu.process(rgb, quality=quality, **params)      # (without a trace: the size-bound path)
    assert got2 == exp_jpg


@needs_ref
@pytest.mark.parametrize("quality,refused", [(83.0, True), (83.1, True), (70.0, True), (83.2, False)])
def test_qualities_below_84_are_refused_like_the_reference(host_emu, quality, refused, capfd):
    """processor.cc:800-806: butteraugli_target > 2.0 -> Process returns false after the message;
    nothing is written.  The rule is on the TARGET, not on the quality: 83.2 interpolates
    (quality.cc:78-85) to 1.9966 and is encoded, by the reference and here."""
    rgb = images.crop(40, 32, 100, 60)
    target = ref._butteraugli_score_for_quality(quality)
    assert (target > 2.0) == refused
    exp = ref.process_params(rgb, target)[0]
    assert (exp is None) == refused
    capfd.readouterr()
    if not refused:
        assert host_emu.process(rgb, quality=quality)[0] == exp
        return
    with pytest.raises(RuntimeError):
        host_emu.process(rgb, quality=quality)
    assert "Guetzli should be called with quality >= 84" in capfd.readouterr().err
    data = _pil_jpeg(rgb, quality=97, subsampling=0)
    assert ref.process_params(data, target)[0] is None
    with pytest.raises(RuntimeError):
        host_emu.process_jpeg(data, quality=quality)


@needs_ref
def test_size_bound_path_with_its_self_checks_in_emulation(host_emu, monkeypatch):
    """GZ_VERIFY_ENTROPY=2: the default (no trace) order of calls -- bound decision, then the scan
    behind the evaluation -- with every candidate coded and checked: bits == the statistics' count,
    bound <= size, device scan == host writer, and no candidate the bound rejects would have won."""
    monkeypatch.setenv("GZ_VERIFY_ENTROPY", "2")
    rgb = images.crop(48, 40, 300, 150)
    exp_jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(95.0))
    got, info = host_emu.process(rgb, quality=95)
    assert got == exp_jpg
    assert info["counters"]["candidates rejected on their size bound"] > 0


@needs_ref
@pytest.mark.parametrize("case", [(48, 40, 300, 150, {}), (136, 88, 60, 20, {}), (61, 43, 10, 30, {}),
                                  (64, 48, 200, 100, dict(force_420=True))])
def test_patched_candidate_planes_equal_a_full_reconstruction_in_emulation(host_emu, monkeypatch, case):
    """gz_config.patch_reconstruct = 2 (GZ_PATCH_RECON): every Compare that relies on the candidate's linear planes
    having been kept current by gz_apply_candidate_steps / gz_apply_coeff_edits checks them against a full
    reconstruction first; ragged sizes patch partial blocks at the right and bottom edges; a 4:2:0 frame never
    patches.  And = 0: no Compare skips its reconstruction.  gz_config.opsin_ahead: the opsin image of those planes
    kept current as well (whole behind the bulk steps, by tiles around the serial steps' edits) and checked likewise.
    Same bytes as the reference."""
    from guetzli_amd import capi
    w, h, x0, y0, kw = case
    L = capi.Library(build_emu.build())
    rgb = images.crop(w, h, x0, y0)
    exp_jpg, _ = ref.process_params(rgb, ref._butteraugli_score_for_quality(95.0), **kw)
    for mode, ahead_env in (("2", "1"), ("2", "0"), ("0", "1")):
        monkeypatch.setenv("GZ_PATCH_RECON", mode)
        monkeypatch.setenv("GZ_OPSIN_AHEAD", ahead_env)
        before = L.compare_counters(all=True)
        got, info = host_emu.process(rgb, quality=95, **kw)
        patched, checked, compares, ahead, ahead_checked = (a - b for a, b in zip(L.compare_counters(all=True), before))
        assert got == exp_jpg
        assert compares > 0 and patched == checked and ahead == ahead_checked and ahead <= patched
        if mode == "2" and not kw:
            assert patched >= info["counters"]["number of iterations"] // 2
            # (the opsin image ahead of the serial steps, its tiles around their edits computed again: behind every
            # iteration's bulk steps -- small images have iterations without any)
            if ahead_env == "0":
                assert ahead == 0
            elif (w, h) == (136, 88):
                assert ahead > patched // 2, (ahead, patched)
        else:
            assert patched == 0 and ahead == 0


@needs_ref
@pytest.mark.parametrize("threads", [0, 1, 3])
def test_code_refresh_helpers_change_nothing_but_the_time(host_emu, monkeypatch, threads):
    """guetzli_amd/host/code_refresh.h: phase B's serial steps with the size model's Huffman codes
    constructed on helper threads -- steps taken ahead of their codes, priced in order when the codes
    arrive, undone beyond the stopping point -- give the reference's bytes, the reference's --verbose
    trace and the same number of coefficient steps as the serial loop (GZ_CODE_THREADS=0), with steps
    really taken ahead and undone."""
    monkeypatch.setenv("GZ_CODE_THREADS", str(threads))
    # (an iteration that follows a short one starts with the reference's own loop and calls the helpers in
    # only after 30 steps -- on this small image every iteration is short: helpers from the first step on)
    monkeypatch.setenv("GZ_CODE_SERIAL_STEPS", "0")
    rgb = images.crop(48, 40, 300, 150)
    target = ref._butteraugli_score_for_quality(95.0)
    exp_jpg, exp_trace = ref.process(rgb, target, want_trace=True)
    got, info = host_emu.process(rgb, quality=95)
    assert got == exp_jpg
    c = info["counters"]
    assert c["phase B code refresh threads"] == threads
    assert (c["phase B steps taken ahead and undone"] > 0) == (threads > 0)
    steps = c["phase B coefficient steps"]
    monkeypatch.setenv("GZ_CODE_THREADS", "0")
    _, info0 = host_emu.process(rgb, quality=95)
    assert info0["counters"]["phase B coefficient steps"] == steps
    monkeypatch.setenv("GZ_CODE_THREADS", str(threads))
    got_t, info_t = host_emu.process(rgb, quality=95, want_trace=True)
    assert got_t == exp_jpg and info_t["trace"] == exp_trace
    # ... and with the switch from the reference's loop to the helpers in the middle of an iteration
    monkeypatch.setenv("GZ_CODE_SERIAL_STEPS", "10")
    got_s, info_s = host_emu.process(rgb, quality=95)
    assert got_s == exp_jpg and info_s["counters"]["phase B coefficient steps"] == steps


@needs_ref
def test_parallel_step_count_of_a_long_prefix_in_emulation(host_emu, monkeypatch):
    """The per-block step counts of a bulk prefix counted by the worker pool (private count arrays per
    range of entries; on the MI355X: the first "up" iteration's 7.5 M entries at 4K), forced onto every
    iteration of a small encode: the reference's bytes."""
    monkeypatch.setenv("GZ_PARALLEL_COUNT_MIN", "1")
    monkeypatch.setenv("GZ_HOST_THREADS", "4")
    rgb = images.crop(48, 40, 300, 150)
    exp_jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(95.0))
    got, info = host_emu.process(rgb, quality=95)
    assert got == exp_jpg
    assert info["counters"]["phase B fast steps"] > 0


@needs_ref
@pytest.mark.parametrize("kw", [{}, {"force_420": True}])
def test_lazy_host_mirror_equals_the_device_image_after_the_search(host_emu, monkeypatch, kw):
    """The host's mirror of the coefficients follows the bulk steps block by block, in whichever direction a
    block lags (also across the turn from "up" to "down"): after every search, with every block caught up, it
    must equal the device image (GZ_CHECK_MIRROR makes the driver check that itself), with and without the
    code-refresh helpers' steps taken ahead and undone."""
    monkeypatch.setenv("GZ_CHECK_MIRROR", "1")
    rgb = images.crop(48, 40, 300, 150)
    for threads in ("0", "3"):
        monkeypatch.setenv("GZ_CODE_THREADS", threads)
        monkeypatch.setenv("GZ_CODE_SERIAL_STEPS", "0")
        exp_jpg, _ = ref.process_params(rgb, ref._butteraugli_score_for_quality(95.0), **kw)
        got, _ = host_emu.process(rgb, quality=95, **kw)
        assert got == exp_jpg
