"""GPU parity tests proper: the hipcc-built gfx950 library, through the C ABI, against the
oracle -- bit-exact on the integer block path AND on the float butteraugli path (the
output JPEG can only be byte-identical if every distance is).  Full-size cases use
size-independent properties where the oracle would take too long."""
import os

import numpy as np
import pytest

import images
import parity_cases as pc
from checkers import assert_bits_equal, oracle, ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import guetzli_amd
    lib = guetzli_amd.load()
    assert lib.device_count() >= 1
    return lib


def test_device_arithmetic_is_ieee_and_uncontracted(L):
    """f32/f64 divide and sqrt correctly rounded, no FMA contraction, RNE conversions."""
    rng = np.random.default_rng(1)
    n = 1 << 16
    a32 = (rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n)).astype(np.float32)
    b32 = (rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n)).astype(np.float32)
    c32 = (rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n)).astype(np.float32)
    a32[:64] = np.float32(1e-41) * np.arange(64, dtype=np.float32)   # denormals
    a64 = rng.standard_normal(n) * 10.0 ** rng.integers(-200, 200, n)
    b64 = rng.standard_normal(n) * 10.0 ** rng.integers(-200, 200, n)
    c64 = rng.standard_normal(n) * 10.0 ** rng.integers(-200, 200, n)
    with np.errstate(all="ignore"):
        assert_bits_equal(L.arith(0, a32, b32), a32 / b32, "f32 divide")
        assert_bits_equal(L.arith(1, np.abs(a32)), np.sqrt(np.abs(a32)), "f32 sqrt")
        assert_bits_equal(L.arith(2, a64, b64), a64 / b64, "f64 divide")
        assert_bits_equal(L.arith(3, np.abs(a64)), np.sqrt(np.abs(a64)), "f64 sqrt")
        assert_bits_equal(L.arith(4, a32, b32, c32), (a32 * b32) + c32, "f32 mul+add unfused")
        assert_bits_equal(L.arith(5, a64, b64, c64), (a64 * b64) + c64, "f64 mul+add unfused")
        small = rng.standard_normal(n) * 10.0 ** rng.integers(-45, 38, n)
        assert_bits_equal(L.arith(6, small), small.astype(np.float32), "f64->f32")


def test_block_kernels(L):
    pc.case_block_kernels(L, n=20000)


@pytest.mark.parametrize("wh", [(444, 258), (61, 43)])
def test_global_order(L, wh):
    pc.case_global_order(L, *wh, x0=0, y0=0)


def test_device_partition_is_std_sort(L, tmp_path):
    """gz_order_partition on the GPU, driven by the product's LazySorted, against std::sort
    itself: every size / tie pattern up to 12M entries (8.4M: the tables of the descent's usual
    kernel are full; 12M: its instantiation with the larger tables), two device thresholds."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_device_order")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread",
                    os.path.join(root, "tests", "cpp", "test_device_order.cc"), "-o", exe, "-ldl"],
                   check=True)
    out = subprocess.run([exe, L.path, "12000017", "16", "65536"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device_order: ok" in out.stdout


def test_device_ranking_is_std_sort(L, tmp_path):
    """k_rank_candidates' per-block std::sort on the GPU (gz_probe_rank_sort) against
    std::sort itself: 1500 arrays of 0..192 keys, tie-heavy and adversarial ones included."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_rank_sort")
    subprocess.run(["g++", "-O2", "-std=c++17", "-DGZ_EMU", "-I" + os.path.join(root, "tests", "emu"),
                    "-I" + os.path.join(root, "guetzli_amd", "csrc"), "-pthread",
                    os.path.join(root, "tests", "cpp", "test_rank_sort.cc"), "-o", exe, "-ldl"],
                   check=True)
    out = subprocess.run([exe, L.path], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device == std::sort" in out.stdout


def test_malta_interior_and_border_tiles(L):
    """k_malta_rolled on an image with interior and border Malta tiles, against the oracle."""
    pc.case_compare(L, 200, 110, x0=100, y0=60, qscales=(5,))


def test_dct_double(L):
    pc.case_dct_double(L, n=20000)


@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (32, 32), (8, 8), (129, 9), (1, 1)])
def test_downsample_component(L, wh):
    pc.case_downsample_component(L, *wh)


def test_dct_double_roundtrip_1080p(L):
    """Full-size property: IDCTDouble(DCTDouble(x)) == x to ~1e-7 relative (the 10-digit
    basis is not exactly orthonormal) and ToFloatPixels of the encoder's coefficients is
    within 1.5 grey levels of the integer IDCT path, over every block of a 1080p plane."""
    rgb = images.tiled(1920, 1080)
    with L.context(rgb, 1.0) as ctx:
        co = ctx.encode_rgb()
        ctx.quantize(None)
        srgb, _ = ctx.reconstruct()
    rng = np.random.default_rng(3)
    blocks = rng.random((32400, 64)) * 255.0 - 128.0
    back = L.dct_double_blocks(L.dct_double_blocks(blocks), inverse=True)
    assert np.abs(back - blocks).max() < 1e-6
    y = L.component_to_float_pixels(co[0], 1920, 1080)
    assert y.shape == (1080, 1920) and np.isfinite(y).all()
    # Y of the integer path: libjpeg colour transform is the identity on Y up to rounding
    yi = (0.299 * srgb[..., 0] + 0.587 * srgb[..., 1] + 0.114 * srgb[..., 2])
    assert np.abs(np.clip(y, 0, 255) - yi).max() < 2.5


@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (32, 32), (8, 8), (129, 9)])
def test_encode_quantize_reconstruct(L, wh):
    pc.case_encode_quantize_reconstruct(L, *wh)


@pytest.mark.parametrize("wh", [(444, 258), (70, 67), (600, 9), (33, 300), (32, 32)])
def test_blur(L, wh):
    pc.case_blur(L, *wh)


def test_paired_row_column_passes(L, monkeypatch):
    """k_blur_h_pk / k_blur_v_pk (the default from 4 MPix on: test_large_odd_image_24_mpix runs
    them at scale) forced at sizes the oracle handles quickly, 16- and 32-row tiles."""
    monkeypatch.setenv("GZ_BLUR_PK", "1")
    pc.case_blur(L, 1100, 300)
    pc.case_stages(L, 440, 250, x0=0, y0=0)
    monkeypatch.setenv("GZ_TILE_ROWS", "32")
    pc.case_blur(L, 840, 200)
    pc.case_compare(L, 444, 258, qscales=(6,))
    monkeypatch.setenv("GZ_BLUR_PK", "0")
    pc.case_blur(L, 840, 200)


@pytest.mark.parametrize("wh", [(256, 192), (72, 48), (35, 41), (444, 258)])
def test_stages(L, wh):
    pc.case_stages(L, *wh, x0=0, y0=0)


def test_config_struct_chooses_instantiations_not_results(L):
    """gz_set_config (round 6: the context's copy of what the GZ_* variables used to say on every call): every
    combination of the chain's instantiation switches on ONE context -- packed / unpacked passes, 16- / 32-row tiles,
    one stream / three, the distance map stored or not -- gives the same distance, block maxima and (where it is
    taken) distance map, bit for bit; and the search loop's form of the evaluation (gz_compare_begin / _end: no
    distance map stored) gives the same distance as gz_compare with the map."""
    rgb = images.crop(600, 264, 40, 20)
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb(download=False)
        ctx.quantize(np.full((3, 64), 5, np.int32), download=False)
        base = ctx.get_config().as_dict()
        if not any(k in os.environ for k in ("GZ_BLUR_PK", "GZ_TILE_ROWS", "GZ_STORE_DISTMAP", "GZ_SINGLE_STREAM")):
            assert base["blur_packed"] == -1 and base["tile_rows"] == 0 and base["store_distmap"] == 0
            assert base["single_stream"] == -1
        d0, dm0, bm0 = ctx.compare()
        ctx.compare_begin()
        assert ctx.compare_end() == d0
        # a second context alive on the device: the default (-1) takes the one-stream chain for both now
        with L.context(images.crop(200, 120, 10, 10), 0.971769) as other:
            other.encode_rgb(download=False)
            other.quantize(np.full((3, 64), 3, np.int32), download=False)
            d, dm, bm = ctx.compare()
            assert d == d0
            pc.assert_bits_equal(dm, dm0, "distance map with a second context alive")
            pc.assert_bits_equal(bm, bm0, "block maxima with a second context alive")
            other.compare()
        for kw in (dict(blur_packed=1, tile_rows=32), dict(blur_packed=0, tile_rows=16), dict(single_stream=1),
                   dict(single_stream=0), dict(blur_packed=1, tile_rows=16, store_distmap=1),
                   dict(side_small=1, malta_pad_bytes=7400)):
            ctx.set_config(**dict(base, **kw))
            d, dm, bm = ctx.compare()
            assert d == d0, kw
            pc.assert_bits_equal(dm, dm0, f"distance map under {kw}")
            pc.assert_bits_equal(bm, bm0, f"block maxima under {kw}")
            ctx.compare_begin()
            assert ctx.compare_end() == d0, kw
            d, _, bm = ctx.compare(want_distmap=False)
            assert d == d0
            pc.assert_bits_equal(bm, bm0, f"block maxima without the map under {kw}")
        with pytest.raises(Exception):
            ctx.set_config(tile_rows=24)


def test_stream_choice_is_made_once_per_compare(L):
    """gz_config.single_stream = -1 looks at the contexts alive on the device; they come and go on other threads WHILE a
    Compare is being enqueued.  One thread evaluates the same candidate 60 times, another creates and destroys contexts
    as fast as it can: every evaluation must give the lone context's distance and block maxima (a fork made for three
    streams that met a join made for one would let k_combine run ahead of the side branches)."""
    import threading
    rgb = images.crop(640, 360, 0, 0)
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb(download=False)
        ctx.quantize(np.full((3, 64), 4, np.int32), download=False)
        d0, _, bm0 = ctx.compare(want_distmap=False)
        stop = threading.Event()
        small = images.crop(64, 48, 5, 5)

        def churn():
            while not stop.is_set():
                with L.context(small, 0.971769) as other:
                    other.encode_rgb(download=False)

        t = threading.Thread(target=churn)
        t.start()
        try:
            for k in range(60):
                d, _, bm = ctx.compare(want_distmap=False)
                assert d == d0, k
                pc.assert_bits_equal(bm, bm0, f"block maxima of evaluation {k} beside context churn")
        finally:
            stop.set()
            t.join()


def test_compare_bees(L):
    """BASELINE config 1 image, full size, three candidate quantisations."""
    pc.case_compare(L, 444, 258, qscales=(1, 2, 6, 14))


def test_compare_small_and_ragged(L):
    pc.case_compare(L, 32, 32, x0=100, y0=100, qscales=(3,))
    pc.case_compare(L, 131, 77, x0=50, y0=20, qscales=(2, 8))


@pytest.mark.skipif(ref is None, reason="oracle/_ref not built")
def test_compare_against_unmodified_reference(L):
    """Same check against the real reference comparator (oracle/_ref), including the
    reference's own quant-matrix search candidates for bees.png."""
    rgb = images.bees()
    h, w, _ = rgb.shape
    target = 0.971769
    rc = ref.comparator(rgb, target)
    with L.context(rgb, target) as ctx:
        co = ctx.encode_rgb()
        assert_bits_equal(co, ref.encode_rgb(rgb), "encode vs reference")
        q = np.ones((3, 64), np.int32)
        q[:, 32:] = 3
        cq = ctx.quantize(q)
        rcq, _, _ = ref.reconstruct(co, w, h, q)
        assert_bits_equal(cq, rcq, "quantize vs reference")
        dist, dm, _ = ctx.compare()
        rdist, rdm = rc.compare(cq)
        assert_bits_equal(dm, rdm, "distmap vs reference")
        assert dist == rdist
    rc.close()


def test_full_size_properties_1080p(L):
    """BASELINE config 2 size (1920x1080).  The oracle needs ~1 s per Compare here, so one
    exact comparison plus size-independent properties: identical candidate => identical
    map (idempotence); q=1 candidate of a tiled image => the map is tile-periodic away from
    the borders; distance == max(map); block maxima consistent."""
    w, h = 1920, 1080
    rgb = images.tiled(w, h)
    target = 0.971769
    with L.context(rgb, target) as ctx:
        co = ctx.encode_rgb()
        q = np.full((3, 64), 4, np.int32)
        cq = ctx.quantize(q)
        d1, m1, b1 = ctx.compare()
        d2, m2, b2 = ctx.compare()
        assert_bits_equal(m1, m2, "idempotence")
        assert d1 == d2 == m1.max()
        pad = np.zeros((ctx.bh * 8, ctx.bw * 8), np.float32)
        pad[:h, :w] = m1
        assert_bits_equal(b1, pad.reshape(ctx.bh, 8, ctx.bw, 8).max(axis=(1, 3)).reshape(-1),
                          "block max")
        oc = oracle.comparator(rgb, target)
        ed, em = oc.compare(cq)
        oc.close()
        assert_bits_equal(m1, em, "1080p distmap vs oracle")
        assert d1 == ed


def test_large_odd_image_24_mpix(L):
    """Beyond 4K, neither dimension a multiple of 4 or 8 (6001 x 4003 = 24 MPix: every kernel's
    generic path at scale, size_t index arithmetic, a 3.9 GB plane arena): the block path, the
    distance map and its block maxima bit for bit against the oracle, and the exact scan size of
    the device entropy coder against the serial host writer."""
    import guetzli_amd
    w, h = 6001, 4003
    rgb = images.tiled(w, h)
    target = 0.971769
    with L.context(rgb, target) as ctx:
        co = ctx.encode_rgb()
        assert_bits_equal(co, oracle.encode_rgb(rgb), "encode_rgb 24 MPix")
        q = np.full((3, 64), 5, np.int32)
        cq = ctx.quantize(q)
        d1, m1, b1 = ctx.compare()
        oc = oracle.comparator(rgb, target)
        ed, em = oc.compare(cq)
        oc.close()
        assert_bits_equal(m1, em, "24 MPix distmap vs oracle")
        assert d1 == ed == m1.max()
        pad = np.zeros((ctx.bh * 8, ctx.bw * 8), np.float32)
        pad[:h, :w] = m1
        assert_bits_equal(b1, pad.reshape(ctx.bh, 8, ctx.bw, 8).max(axis=(1, 3)).reshape(-1), "block max")
        host = guetzli_amd.load_host()
        counts = ctx.jpeg_histograms(q)
        head, depth, code = host.jpeg_head(counts, w, h, q)
        n = ctx.jpeg_scan(3, depth, code)
        exp = host.write_jpeg(cq, w, h, q)
        assert len(head) + n + 2 == len(exp)
        assert head + ctx.jpeg_scan_bytes(cap=n + 16) + b"\xff\xd9" == exp


def test_huge_image_256_mpix_periodicity(L):
    """16384 x 16384 (268 MPix, a 44 GB plane arena, 4.2 M blocks): far beyond what the oracle
    can check in a test, so size-independent properties.  The image is bees.png tiled from the
    origin; with the 8x8 block grid its content repeats every (888, 1032) pixels, so the distance
    map of a uniformly quantised candidate must repeat with that period away from the borders
    (bit for bit: every sample sees identical inputs through identical operations), and must
    equal the same window of the 3840x2160 image's map (a size whose kernels are pinned to the
    oracle by the tests above).  An index that wrapped at 2^31 anywhere would break both.  Also:
    distance == max(map), block maxima consistent, the device entropy coder's exact size
    against the serial host writer."""
    import guetzli_amd
    w = h = 16384
    target = 0.971769
    q = np.full((3, 64), 5, np.int32)
    px, py = 888, 1032   # lcm(444, 8), lcm(258, 8)
    with L.context(images.tiled(3840, 2160), target) as small:
        small.encode_rgb(download=False)
        small.quantize(q, download=False)
        _, ms, _ = small.compare(want_block_max=False)
    win_small = ms[py:2 * py, px:2 * px].copy()   # >= 96 px from every border of the 4K image
    del ms
    rgb = images.tiled(w, h)
    with L.context(rgb, target) as ctx:
        ctx.encode_rgb(download=False)
        cq = ctx.quantize(q)
        d1, m1, b1 = ctx.compare()
        assert d1 == m1.max()
        ref_win = m1[py:2 * py, px:2 * px]
        assert_bits_equal(ref_win, win_small, "16K window vs the same window of the 4K image")
        for ky, kx in ((0, 1), (1, 0), (7, 9), (13, 16), (13, 1), (1, 16)):
            y0, x0 = py * (1 + ky), px * (1 + kx)
            assert y0 + py <= h - 96 and x0 + px <= w - 96
            assert_bits_equal(m1[y0:y0 + py, x0:x0 + px], ref_win, f"period ({ky}, {kx})")
        assert_bits_equal(b1, m1.reshape(ctx.bh, 8, ctx.bw, 8).max(axis=(1, 3)).reshape(-1), "block max")
        del m1
        host = guetzli_amd.load_host()
        counts = ctx.jpeg_histograms(q)
        head, depth, code = host.jpeg_head(counts, w, h, q)
        n = ctx.jpeg_scan(3, depth, code)
        exp = host.write_jpeg(cq, w, h, q)
        assert len(head) + n + 2 == len(exp)
        got = ctx.jpeg_scan_bytes(cap=n + 16)
        assert head + got + b"\xff\xd9" == exp


def test_block_search_small_ragged(L):
    pc.case_block_search(L, 45, 27)
    pc.case_block_search(L, 64, 40, x0=10, y0=10, qs=2)


def test_compare_blocks(L):
    pc.case_compare_blocks(L, 45, 27, x0=100, y0=60)
    pc.case_compare_blocks(L, 200, 120, x0=100, y0=60, n=200)


def test_block_search_bees(L):
    """Phase A on the whole BASELINE config-1 image (1848 blocks, ~300k CompareBlock
    evaluations) against the oracle, bit for bit."""
    pc.case_block_search(L, 444, 258, x0=0, y0=0, qs=2)


@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (32, 32), (129, 9), (8, 8)])
def test_jpeg_entropy(L, wh):
    """Device symbol statistics + device scan vs the reference's WriteJpeg, byte for byte."""
    import guetzli_amd
    pc.case_jpeg_entropy(L, guetzli_amd.load_host(), *wh, check_histograms=wh[0] * wh[1] < 20000)


def test_jpeg_entropy_1080p():
    """Full-size scan (32 400 MCUs): byte-identical to the serial host writer, which the
    CPU suite pins to the reference."""
    import guetzli_amd
    Lb = guetzli_amd.load()
    pc.case_jpeg_entropy(Lb, guetzli_amd.load_host(), 1920, 1080, check_histograms=False)


# SHA-256 of the reference's output JPEGs (unmodified guetzli, default flags = quality 95),
# BASELINE.md section 2; the bees hash is re-derived from oracle/_ref in the CPU suite.
GOLDEN_JPEG_SHA = {
    (444, 258, 95): "f2673f12a4856e020627fa151493a80b1cb2ee4dc81e28afc62dc089baf50242",
    (444, 258, 84): "95f509f457ce8ddd85087c804664539e0ef7b1f3a6c6ca1cbedcd9ba29a89379",
    (1920, 1080, 95): "9c0eb414b8e73f4372c0b089eafe2350e6ff2ae83926d1c0c5f35cb5f7919729",
    (3840, 2160, 95): "481507d21e4d37f296ae6a2a93a84d950c4a135df310b3408a64390b25d59c05",
    (3840, 2160, 84): "f3be1e4385a977853f7cd224e1722a728c1fa0bc68f1115c27e657c23a028ac0",
}
GOLDEN_TRACE_SHA = {
    (444, 258, 95): "954ec7623366bc3c345fc5b0748017f9a5e0128aba0917a249cca390a615f787",
}


@pytest.mark.parametrize("q", [95, 84])
def test_whole_encode_bees_bit_identical_jpeg(q):
    """BASELINE config 1: guetzli tests/bees.png --quality 95 (and 84): the output JPEG is
    byte-identical to the reference's, and so is the --verbose trace."""
    import hashlib
    import guetzli_amd
    rgb = images.bees()
    jpg, info = guetzli_amd.process(rgb, quality=q, want_trace=True)
    assert hashlib.sha256(jpg).hexdigest() == GOLDEN_JPEG_SHA[(444, 258, q)]
    if (444, 258, q) in GOLDEN_TRACE_SHA:
        assert hashlib.sha256(info["trace"].encode()).hexdigest() == GOLDEN_TRACE_SHA[(444, 258, q)]


def test_memory_pool_reuse_and_trim_do_not_change_results(L):
    """Contexts of one size reuse the cached device blocks, streams and events of the ones
    destroyed before them (stale contents, not zeroes); gz_trim_pool releases the cache.  The
    JPEG is the same every time."""
    import hashlib
    import guetzli_amd
    rgb = images.bees()
    first, _ = guetzli_amd.process(rgb, quality=95)
    again, _ = guetzli_amd.process(rgb, quality=90)      # same sizes, different content in the blocks
    assert L.lib.gz_trim_pool() == 0
    fresh, _ = guetzli_amd.process(rgb, quality=95)      # allocates anew
    reused, _ = guetzli_amd.process(rgb, quality=95)     # from the pool
    assert first == fresh == reused
    assert hashlib.sha256(first).hexdigest() == GOLDEN_JPEG_SHA[(444, 258, 95)]
    assert again != first


def test_png_file_in_jpeg_out_matches_the_reference_golden():
    """`guetzli tests/bees.png out.jpg` end to end: the PNG bytes go through the product's own
    reader (png_reader.cc) and the JPEG is the reference's."""
    import hashlib
    import guetzli_amd
    jpg, _ = guetzli_amd.process_png(open(images.BEES, "rb").read(), quality=95)
    assert hashlib.sha256(jpg).hexdigest() == GOLDEN_JPEG_SHA[(444, 258, 95)]


def test_whole_encode_1080p_bit_identical_jpeg():
    """BASELINE config 2 (1920x1080, quality 95): byte-identical to the reference output."""
    import hashlib
    import guetzli_amd
    rgb = images.tiled(1920, 1080)
    jpg, info = guetzli_amd.process(rgb, quality=95)
    assert len(jpg) == 721187
    assert hashlib.sha256(jpg).hexdigest() == GOLDEN_JPEG_SHA[(1920, 1080, 95)]


@pytest.mark.parametrize("q,size", [(95, 2895866), (84, 1430837)])
def test_whole_encode_4k_bit_identical_jpeg(q, size):
    """BASELINE configs 3 and 4 (3840x2160, quality 95 and 84): byte-identical to the
    reference output."""
    import hashlib
    import guetzli_amd
    rgb = images.tiled(3840, 2160)
    jpg, info = guetzli_amd.process(rgb, quality=q)
    assert len(jpg) == size
    assert hashlib.sha256(jpg).hexdigest() == GOLDEN_JPEG_SHA[(3840, 2160, q)]


@pytest.mark.parametrize("mode", ["2", "0"])
def test_patched_candidate_planes_equal_a_full_reconstruction_1080p(L, monkeypatch, mode):
    """gz_config.patch_reconstruct: the calls that change the candidate (bulk steps, serial steps' edits) transform
    the block positions they touch again, and the Compare behind them skips its full reconstruction.  Mode 2 checks
    the patched planes against a full reconstruction before EVERY such Compare (a difference fails the encode);
    mode 0 is the chain with its reconstruction in front.  Same bytes as the reference either way."""
    import hashlib
    import guetzli_amd
    monkeypatch.setenv("GZ_PATCH_RECON", mode)
    before = L.compare_counters(all=True)
    jpg, info = guetzli_amd.process(images.tiled(1920, 1080), quality=95)
    patched, checked, compares, ahead, ahead_checked = (a - b for a, b in zip(L.compare_counters(all=True), before))
    assert hashlib.sha256(jpg).hexdigest() == GOLDEN_JPEG_SHA[(1920, 1080, 95)]
    assert compares >= info["counters"]["number of iterations"]
    if mode == "2":
        assert patched == checked and patched > 0.8 * info["counters"]["number of iterations"]
        # gz_config.opsin_ahead: the opsin image of those planes was in place as well (and checked) wherever an
        # iteration had bulk steps
        # (not under a forced one-stream chain or with the switch off: the suite also runs in those modes)
        if os.environ.get("GZ_SINGLE_STREAM") == "1" or os.environ.get("GZ_OPSIN_AHEAD") == "0":
            assert ahead == 0
        else:
            assert ahead == ahead_checked and ahead > 0.6 * info["counters"]["number of iterations"]
    else:
        assert patched == 0 and checked == 0 and ahead == 0


@pytest.mark.parametrize("level", ["1", "2"])
def test_whole_encode_with_self_checks_1080p(monkeypatch, level):
    """GZ_VERIFY_ENTROPY=1: every candidate's device scan equals the host writer's bytes and
    the host mirror of the image equals the device image after every iteration.  =2: the same
    checks on the DEFAULT path's order of calls -- the size-bound decision first, the scan late,
    behind the evaluation and the next order's construction; a candidate the bound rejects is coded
    anyway and must lose with its real size."""
    import hashlib
    import guetzli_amd
    monkeypatch.setenv("GZ_VERIFY_ENTROPY", level)
    jpg, info = guetzli_amd.process(images.tiled(1920, 1080), quality=95)
    assert hashlib.sha256(jpg).hexdigest() == GOLDEN_JPEG_SHA[(1920, 1080, 95)]
    if level == "2":
        assert info["counters"]["candidates rejected on their size bound"] > 100


def test_concurrent_encodes_on_one_gpu_are_deterministic():
    """Batch mode (guetzli_amd.batch.encode_concurrent): several images in flight on one GPU,
    one host thread and one device context each, give exactly the bytes of encoding them one
    after the other."""
    import guetzli_amd
    from guetzli_amd.batch import encode_concurrent
    host = guetzli_amd.load_host()
    imgs = [images.shifted(images.bees(), k) for k in range(6)]
    proc = lambda im: host.process(im, quality=95)
    seq = [proc(im)[0] for im in imgs]
    con = [j for j, _ in encode_concurrent(imgs, proc, workers=3)]
    assert con == seq
    assert len(set(seq)) == len(seq)   # they really are different images


def _golden_cases():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_hashes.json")
    return sorted(json.load(open(path)).items())


@pytest.mark.parametrize("key,exp", _golden_cases())
def test_whole_encode_golden_hashes(key, exp):
    """Whole encodes against hashes the UNMODIFIED reference produced (tools/gen_golden_hashes.py,
    minutes of CPU each): odd sizes (width not a multiple of 4 or 8 -> the kernels' generic
    paths), other qualities, and a synthetic image with statistics unlike bees.png."""
    import hashlib
    import guetzli_amd
    kind, size, q = key.split("_")
    w, h = (int(v) for v in size.split("x"))
    rgb = images.tiled(w, h) if kind == "tiled" else images.synthetic(w, h)
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == exp["rgb_sha256"]
    jpg, _ = guetzli_amd.process(rgb, quality=float(q[1:]))
    assert len(jpg) == exp["bytes"]
    assert hashlib.sha256(jpg).hexdigest() == exp["jpeg_sha256"]


@pytest.mark.parametrize("wh", [(24, 40), (31, 64), (1, 1)])
def test_small_images_emit_the_unquantised_jpeg(wh):
    """w or h < 32 (no butteraugli, processor.cc:832-838): the forward transform runs on the
    device through gz_encode_rgb_only; bytes equal the reference's when oracle/_ref is present,
    else the oracle restatement's coefficients through the host writer."""
    import guetzli_amd
    w, h = wh
    rgb = images.crop(w, h, 200, 100)
    jpg, _ = guetzli_amd.process(rgb, quality=95)
    if ref is not None:
        exp, _ = ref.process(rgb, ref._butteraugli_score_for_quality(95.0))
        assert jpg == exp
    host = guetzli_amd.load_host()
    assert jpg == host.write_jpeg(oracle.encode_rgb(rgb), w, h, None)


def _jpeg_input_cases():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_input_hashes.json")
    return sorted(json.load(open(path)).items()) if os.path.exists(path) else []


@pytest.mark.parametrize("name,exp", _jpeg_input_cases())
def test_jpeg_input_golden_hashes(name, exp):
    """guetzli::Process(params, stats, jpeg_data, &out): hashes from the unmodified reference
    (tools/gen_golden_hashes.py).  The input stream is written by Pillow here; if this box's
    libjpeg writes it differently the case cannot be compared and is skipped."""
    import hashlib
    import io
    from PIL import Image
    import guetzli_amd
    kw = dict(exp["pil"])
    if "comment" in kw:
        kw["comment"] = kw["comment"].encode()
    b = io.BytesIO()
    Image.fromarray(images.tiled(exp["w"], exp["h"])).save(b, "JPEG", **kw)
    data = b.getvalue() + (b"" if exp["clear_metadata"] else b"TAIL")
    if hashlib.sha256(data).hexdigest() != exp["input_sha256"]:
        pytest.skip("Pillow writes a different input stream on this machine")
    host = guetzli_amd.load_host()
    jpg, _ = host.process_jpeg(data, quality=exp["quality"], clear_metadata=exp["clear_metadata"])
    assert len(jpg) == exp["bytes"]
    assert hashlib.sha256(jpg).hexdigest() == exp["jpeg_sha256"]


# ------------------------------------------------------------------ YUV 4:2:0 (row f4) --
needs_ref = pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")


@pytest.mark.parametrize("wh", [(48, 40), (33, 35), (47, 31), (444, 258), (129, 70)])
def test_frame420(L, wh):
    """OutputImage::Downsample with PreProcessChannel, the 2x2 pixel model (against the
    reference image after shuffled update histories), Compare and the 16x16 block weights."""
    pc.case_frame420(L, *wh, oracle)


def test_frame420_preprocessing_branches(L):
    rgb = pc.colourful(200, 136)
    pc.case_frame420(L, 200, 136, oracle, rgb=rgb)


def test_block_search420(L):
    pc.case_block_search420(L, 45, 27, oracle, x0=100, y0=60)
    pc.case_block_search420(L, 130, 75, oracle, x0=10, y0=10, qs=2)
    pc.case_block_search420(L, 64, 48, oracle, x0=10, y0=10, qs=2, lookahead=2, new_model=False)


def test_block_search_masks_and_params(L):
    """Params::zeroing_greedy_lookahead in {1, 2, 5}, new_zeroing_model = false and every
    component mask on a 4:4:4 frame (processor.cc:364-467)."""
    pc.case_block_search_masks444(L, 96, 64, oracle, x0=100, y0=60)
    pc.case_block_search_masks444(L, 61, 43, oracle, x0=50, y0=60, lookahead=1)
    pc.case_block_search_masks444(L, 61, 43, oracle, x0=50, y0=60, lookahead=2)
    pc.case_block_search_masks444(L, 61, 43, oracle, x0=50, y0=60, lookahead=5, new_model=False)


@needs_ref
@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (32, 32), (129, 9)])
def test_jpeg_entropy420(L, wh):
    import guetzli_amd
    pc.case_jpeg_entropy420(L, guetzli_amd.load_host(), *wh, ref)


@pytest.mark.parametrize("wh", [(444, 258), (1021, 515), (1920, 1080)])
def test_patched_candidate_planes(L, wh):
    pc.case_patched_candidate_planes(L, *wh)


def test_global_order420(L):
    pc.case_global_order420(L, 130, 75, oracle, x0=100, y0=60)


def _params_cases():
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "golden", "params_hashes.json")
    cases = dict(json.load(open(path))) if os.path.exists(path) else {}
    # round 3: the 4:2:0 path, another quality, an odd size and 4:2:0 JPEG input at BASELINE sizes
    # (tools/gen_golden_hashes_r3.py, one file per case)
    r3 = os.path.join(here, "golden", "params_r3")
    if os.path.isdir(r3):
        for f in sorted(os.listdir(r3)):
            if f.endswith(".json"):
                cases[f[:-5]] = json.load(open(os.path.join(r3, f)))
    return sorted(cases.items())


@pytest.mark.parametrize("name,exp", _params_cases())
def test_whole_encode_params_golden_hashes(name, exp, monkeypatch):
    """Whole encodes with non-default guetzli::Params -- try_420 / force_420, zeroing
    look-ahead, the old zeroing model, use_silver_screen -- greyscale input and YUV 4:2:0 JPEG
    input, against hashes the UNMODIFIED reference produced (tools/gen_golden_hashes_r2.py)."""
    import hashlib
    import io
    from PIL import Image
    import guetzli_amd
    kind, w, h = exp["image"]
    if kind == "bees":
        rgb = images.bees()
    elif kind == "grey":
        rgb = np.repeat(images.tiled(w, h)[:, :, 1:2], 3, axis=2).copy()
    else:
        rgb = images.tiled(w, h) if kind == "tiled" else images.synthetic(w, h)
    params = dict(exp["params"])
    if "silver" in params:
        params["use_silver_screen"] = params.pop("silver")
    host = guetzli_amd.load_host()
    if "pil" in exp:
        kw = dict(exp["pil"])
        if "comment" in kw:
            kw["comment"] = kw["comment"].encode()
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", **kw)
        data = b.getvalue() + (b"" if params.get("clear_metadata", True) else b"TAIL")
        if hashlib.sha256(data).hexdigest() != exp["input_sha256"]:
            pytest.skip("Pillow writes a different input stream on this machine")
        jpg, _ = host.process_jpeg(data, quality=exp["quality"], **params)
    else:
        assert hashlib.sha256(rgb.tobytes()).hexdigest() == exp["rgb_sha256"]
        jpg, _ = host.process(rgb, quality=exp["quality"], **params)
    assert len(jpg) == exp["bytes"]
    assert hashlib.sha256(jpg).hexdigest() == exp["jpeg_sha256"]


def _degenerate_cases():
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degenerate")
    if not os.path.isdir(d):
        return []
    return [(f[:-5], json.load(open(os.path.join(d, f)))) for f in sorted(os.listdir(d)) if f.endswith(".json")]


def degenerate_input(exp):
    """(rgb as the reference's front end hands it to Process, params) of a tests/golden/degenerate case."""
    import hashlib
    import os
    import guetzli_amd
    if "png" in exp:
        here = os.path.dirname(os.path.abspath(__file__))
        data = open(os.path.join(here, "golden", "degenerate", exp["png"]), "rb").read()
        assert hashlib.sha256(data).hexdigest() == exp["png_sha256"]
        rgb = guetzli_amd.read_png(data)          # the product's ReadPNG (host/png_reader.cc)
    else:
        spec = exp["image"]
        kind, w, h = spec[:3]
        rgb = {"flat": lambda: images.flat(w, h, spec[3]), "stripes": lambda: images.stripes(w, h),
               "noise": lambda: images.noise(w, h), "tiled": lambda: images.tiled(w, h)}[kind]()
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == exp["rgb_sha256"]
    return rgb, dict(exp["params"])


@pytest.mark.parametrize("name,exp", _degenerate_cases())
def test_degenerate_content_golden_hashes(name, exp):
    """Content at the edges of the search, against hashes the UNMODIFIED reference produced
    (tools/gen_goldens.py degenerate): flat black / white / grey / red (all AC zero; the v < 1e-4
    branch of CalculateDiffmap, butteraugli.cc:722-732; try_420 / force_420 on flat chroma),
    saturated-primary stripes, uniform noise (189-candidate zeroing lists), 33- and 32-pixel
    slivers, and RGBA / 16-bit / interlaced grey+alpha / palette+tRNS PNG files through the
    product's ReadPNG and the full encode."""
    import hashlib
    import guetzli_amd
    rgb, params = degenerate_input(exp)
    jpg, _ = guetzli_amd.load_host().process(rgb, quality=exp["quality"], **params)
    assert len(jpg) == exp["bytes"]
    assert hashlib.sha256(jpg).hexdigest() == exp["jpeg_sha256"]


def _photo_cases():
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "photos")
    if not os.path.isdir(d):
        return []
    return [(f[:-5], json.load(open(os.path.join(d, f)))) for f in sorted(os.listdir(d)) if f.endswith(".json")]


def photo_input(exp):
    """What a tests/golden/photos case hands to Process: (rgb, None) or (None, jpeg bytes)."""
    import hashlib
    if "jpeg_input" in exp:
        data = images.photo_bytes(exp["jpeg_input"].split(".")[0].replace("hubble_deep_field", "hubble"))
        assert hashlib.sha256(data).hexdigest() == exp["input_sha256"]
        return None, data
    spec = exp["image"]
    kind, w, h = spec[:3]
    rgb = {"photo": lambda: images.photo(spec[3]), "mosaic": lambda: images.mosaic(w, h),
           "bees": images.bees}[kind]()
    assert rgb.shape == (h, w, 3)
    if hashlib.sha256(rgb.tobytes()).hexdigest() != exp["rgb_sha256"]:
        pytest.skip("this machine's JPEG decoder gives other pixels for the committed photograph")
    return rgb, None


@pytest.mark.parametrize("name,exp", _photo_cases())
def test_photo_golden_hashes(name, exp, capfd):
    """Round 5: real photographs -- skin, sky gradients, wood grain, fur,
    gravel, a star field, a fundus image; as RGB at q95 / q84 / a fractional quality, as the
    camera's own JPEG stream (4:4:4 and 4:2:0, with and without metadata, with try_420) -- bees.png
    at q100 / q99 / q97.5 / q85.5 / q110 and the refused q83 (processor.cc:800-806), and a
    3840x2160 / 1920x1080 mosaic of the photographs without any period, against hashes the
    UNMODIFIED reference produced (tools/gen_goldens.py photos; licences in
    tests/golden/photos/LICENSES.md)."""
    import hashlib
    import guetzli_amd
    rgb, data = photo_input(exp)
    params = dict(exp["params"])
    host = guetzli_amd.load_host()
    if exp.get("refused"):
        with pytest.raises(RuntimeError):
            host.process(rgb, quality=exp["quality"], **params)
        assert "quality >= 84" in capfd.readouterr().err
        return
    if data is not None:
        jpg, _ = host.process_jpeg(data, quality=exp["quality"], **params)
    else:
        jpg, _ = host.process(rgb, quality=exp["quality"], **params)
    assert len(jpg) == exp["bytes"]
    assert hashlib.sha256(jpg).hexdigest() == exp["jpeg_sha256"]


def test_config5_all_64_reference_hashes():
    """BASELINE configs[4], the whole batch of an 8-GPU run on this one GPU: the 64 3840x2160
    images (the bench image circularly shifted by (37k, 53k)) through guetzli_amd.batch.run_config5,
    4 in flight, every output against the hash the UNMODIFIED reference produced for it
    (tests/golden/config5/k0..k63.json, ~20 CPU-minutes of reference time each;
    /root/reference/tests/golden_test.sh:24-26 is the reference's form of the same check)."""
    import json
    import os
    import guetzli_amd
    from guetzli_amd.batch import run_config5
    here = os.path.dirname(os.path.abspath(__file__))
    gold = {}
    for k in range(64):
        r = json.load(open(os.path.join(here, "golden", "config5", f"k{k}.json")))
        assert (r["k"], r["w"], r["h"], r["quality"]) == (k, 3840, 2160, 95.0)
        gold[k] = r
    base = images.tiled(3840, 2160)
    host = guetzli_amd.load_host()
    recs, seconds = run_config5(lambda k: images.shifted(base, k), 64,
                                lambda rgb: host.process(rgb, quality=95), workers=4)
    assert [r["index"] for r in recs] == list(range(64))
    bad = [r["index"] for r in recs
           if (r["bytes"], r["sha256"]) != (gold[r["index"]]["bytes"], gold[r["index"]]["jpeg_sha256"])]
    assert not bad, f"config-5 images with bytes unlike the reference's: {bad}"
    print(f"config 5: 64 of 64 outputs equal to the reference's, {seconds:.1f} s = "
          f"{64 * 3840 * 2160 / 1e6 / seconds:.1f} MPix/s on one GPU")


def test_config5_members_from_png_bytes():
    """The same batch handed over as PNG files (what `guetzli in.png out.jpg` reads): two members
    through process_png give the reference's bytes."""
    import hashlib
    import io
    import json
    import os
    from PIL import Image
    import guetzli_amd
    here = os.path.dirname(os.path.abspath(__file__))
    base = images.tiled(3840, 2160)
    for k in (5, 41):
        b = io.BytesIO()
        Image.fromarray(images.shifted(base, k)).save(b, "PNG", compress_level=1)
        jpg, _ = guetzli_amd.process_png(b.getvalue(), quality=95)
        r = json.load(open(os.path.join(here, "golden", "config5", f"k{k}.json")))
        assert (len(jpg), hashlib.sha256(jpg).hexdigest()) == (r["bytes"], r["jpeg_sha256"])


def test_contexts_on_a_foreign_current_device_are_safe(L):
    """Every entry point runs on its context's device and restores the caller's:
    with one GPU this checks at least that nothing changes the current device."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    dev = ctypes.c_int(-1)
    rgb = images.crop(64, 48)
    with L.context(rgb, 1.0) as ctx:
        ctx.encode_rgb()
        ctx.quantize(None)
        ctx.compare()
        assert hip.hipGetDevice(ctypes.byref(dev)) == 0 and dev.value == 0


def test_large_image_beyond_the_device_descent_reference_hash():
    """7680x4320 (four times BASELINE's largest size): an iteration's order has 12.5 M entries, more
    than the device's quick-select descent takes by itself (8.4 M: its workgroups' tables), so the
    first partitions of every iteration are driven from the host (gz_order_partition) before the
    descent takes over -- the output must still be the unmodified reference's
    (tests/golden/large/: 75 CPU-minutes of reference time)."""
    import glob
    import hashlib
    import json
    import os
    import guetzli_amd
    here = os.path.dirname(os.path.abspath(__file__))
    cases = sorted(glob.glob(os.path.join(here, "golden", "large", "*.json")))
    if not cases:
        pytest.skip("no fixture under tests/golden/large")
    for path in cases:
        exp = json.load(open(path))
        kind, w, h = exp["image"][:3]
        assert kind == "tiled" and not exp["params"]
        rgb = images.tiled(w, h)
        assert hashlib.sha256(rgb.tobytes()).hexdigest() == exp["rgb_sha256"]
        jpg, info = guetzli_amd.process(rgb, quality=exp["quality"])
        assert len(jpg) == exp["bytes"], (path, len(jpg), exp["bytes"])
        assert hashlib.sha256(jpg).hexdigest() == exp["jpeg_sha256"], path
        assert info["counters"]["phase B device partitions"] > 0   # (the host-driven ones)
