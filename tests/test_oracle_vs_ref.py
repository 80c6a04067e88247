"""Pins the CPU restatement (oracle/gz_oracle.cc) against the UNMODIFIED reference
(oracle/_ref/libgz_ref.so): every stage of the hot path, bit for bit.

CPU-only.  Skipped when the prebuilt reference library is absent (it can only be
built where /root/reference exists; the built file travels with the snapshot); the committed
fixtures under tests/golden/ -- the reference's JPEG hashes, read by tests/test_gpu_parity.py --
do not need it."""
import numpy as np
import pytest

import images
from checkers import assert_bits_equal, oracle, ref

pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")

RNG = np.random.default_rng(20260921)


def rand_blocks(n, lim):
    b = RNG.integers(-lim, lim + 1, size=(n, 64)).astype(np.int16)
    # sparsify half of them like real quantised data
    b[: n // 2][RNG.random((n // 2, 64)) < 0.7] = 0
    return b


def test_idct_blocks():
    blocks = np.concatenate([
        rand_blocks(3000, 4096), rand_blocks(1000, 300), rand_blocks(200, 32767),
        np.full((1, 64), 32767, np.int16), np.full((1, 64), -32768, np.int16),
        np.zeros((1, 64), np.int16)])
    for b in blocks:
        assert_bits_equal(oracle.idct_block(b), ref.idct_block(b), "idct")


def test_fdct_blocks():
    blocks = np.concatenate([
        RNG.integers(-128, 128, size=(3000, 64)).astype(np.int16),
        np.full((1, 64), -128, np.int16), np.full((1, 64), 127, np.int16),
        (RNG.integers(0, 2, size=(200, 64)) * 255 - 128).astype(np.int16)])
    for b in blocks:
        assert_bits_equal(oracle.fdct_block(b), ref.fdct_block(b), "fdct")


def test_quantize_blocks():
    for _ in range(2000):
        b = rand_blocks(1, 4096)[0]
        q = RNG.integers(1, 64, size=64).astype(np.int32)
        ob, oc = oracle.quantize_block(b, q)
        rb, rc = ref.quantize_block(b, q)
        assert_bits_equal(ob, rb, "quantize")
        assert oc == rc


def test_dct_double_blocks():
    """dct_double.cc:76-85."""
    blocks = np.concatenate([
        RNG.integers(-2048, 2048, size=(1500, 64)).astype(np.float64),
        RNG.random((500, 64)) * 255.0,
        RNG.standard_normal((500, 64)) * 10.0 ** RNG.integers(-30, 30, (500, 64)),
        np.zeros((1, 64)), -np.zeros((1, 64))])
    for b in blocks:
        assert_bits_equal(oracle.dct_double(b), ref.dct_double(b), "dct_double")
        assert_bits_equal(oracle.dct_double(b, True), ref.dct_double(b, True), "idct_double")


@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (32, 32), (8, 8), (1, 1), (17, 9)])
def test_to_float_pixels_and_downsample(wh):
    """OutputImageComponent::ToFloatPixels and OutputImage::Downsample without
    sharpen/blur (= SetDownsampledCoefficients), output_image.cc:99-121,265-340."""
    w, h = wh
    co = ref.encode_rgb(images.tiled(w + 30, h + 40)[40:, 30:].copy())
    for c in range(3):
        assert_bits_equal(oracle.to_float_pixels(co[c], w, h), ref.to_float_pixels(co[c], w, h),
                          "ToFloatPixels")
    for fx, fy in ((2, 2),):   # the only subsampling UpdatePixelsForBlock supports besides 1x1
        ou, ov = oracle.downsample_chroma(co, w, h, fx, fy)
        ru, rv = ref.downsample_chroma(co, w, h, fx, fy)
        assert_bits_equal(ou, ru, f"downsample U {fx}x{fy}")
        assert_bits_equal(ov, rv, f"downsample V {fx}x{fy}")


def test_color_and_gamma_tables():
    px = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(256), np.arange(256),
                              indexing="ij"), -1).reshape(-1, 3).astype(np.uint8)
    assert_bits_equal(oracle.ycbcr_to_rgb(px), ref.ycbcr_to_rgb(px), "ycbcr")
    assert_bits_equal(oracle.srgb_table(), ref.srgb_table(), "srgb lut")


@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (32, 32), (8, 8), (1, 1), (17, 9)])
def test_encode_rgb(wh):
    w, h = wh
    rgb = images.crop(w, h, 13, 7)
    assert_bits_equal(oracle.encode_rgb(rgb), ref.encode_rgb(rgb), "encode_rgb")
    noise = RNG.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    assert_bits_equal(oracle.encode_rgb(noise), ref.encode_rgb(noise), "encode_rgb noise")


@pytest.mark.parametrize("wh", [(444, 258), (61, 43), (33, 40)])
def test_reconstruct(wh):
    w, h = wh
    rgb = images.crop(w, h)
    co = ref.encode_rgb(rgb)
    q = np.stack([RNG.integers(1, 12, size=64), RNG.integers(1, 20, size=64),
                  RNG.integers(1, 20, size=64)]).astype(np.int32)
    for qq in (None, q):
        o = oracle.reconstruct(co, w, h, qq)
        r = ref.reconstruct(co, w, h, qq)
        for a, b, nm in zip(o, r, ("coeffs", "srgb", "linear")):
            assert_bits_equal(a, b, f"reconstruct {nm} q={qq is not None}")


SIGMAS = [1.2, 7.46953768697, 3.734768843485, 1.8673844217425, 10.6666499623,
          9.24456601467, 2.3770330432, 9.04353323561, 1.72547472444]


def test_compute_kernel():
    for s in SIGMAS:
        assert_bits_equal(oracle.compute_kernel(s), ref.compute_kernel(s), f"kernel {s}")


@pytest.mark.parametrize("wh", [(97, 61), (32, 32), (8, 8), (50, 33)])
def test_blur(wh):
    w, h = wh
    plane = (RNG.random((h, w)) * 255).astype(np.float32)
    for s in SIGMAS:
        if wh == (8, 8) and s > 2:
            continue   # the reference indexes out of bounds when size < radius
        for br in (0.0, 1.0, -0.0724948220913, 0.147068973249):
            assert_bits_equal(oracle.blur(plane, s, br), ref.blur(plane, s, br),
                              f"blur s={s} br={br}")


def linear_pair(w, h, x0=0, y0=0, qscale=6):
    rgb = images.crop(w, h, x0, y0)
    co = ref.encode_rgb(rgb)
    q = np.full((3, 64), qscale, np.int32)
    _, _, lin1 = ref.reconstruct(co, w, h, q)
    _, _, lin0 = ref.reconstruct(co, w, h, None)
    lut = ref.srgb_table()
    lin_orig = lut[rgb].astype(np.float32).transpose(2, 0, 1).copy()
    return rgb, co, lin_orig, lin1


@pytest.mark.parametrize("wh", [(120, 72), (8, 8), (33, 47)])
def test_opsin(wh):
    w, h = wh
    _, _, lin0, lin1 = linear_pair(w, h, 100, 60)
    for lin in (lin0, lin1):
        assert_bits_equal(oracle.opsin(lin), ref.opsin(lin), "opsin")
    noise = (RNG.random((3, h, w)) * 255).astype(np.float32)
    assert_bits_equal(oracle.opsin(noise), ref.opsin(noise), "opsin noise")


@pytest.mark.parametrize("wh", [(120, 72), (47, 33)])
def test_separate_frequencies(wh):
    w, h = wh
    _, _, lin0, lin1 = linear_pair(w, h, 40, 100)
    for lin in (lin0, lin1):
        xyb = ref.opsin(lin)
        assert_bits_equal(oracle.separate_frequencies(xyb), ref.separate_frequencies(xyb),
                          "separate_frequencies")


@pytest.mark.parametrize("wh", [(120, 72), (47, 33)])
def test_mask(wh):
    w, h = wh
    _, _, lin0, lin1 = linear_pair(w, h, 200, 30)
    x0, x1 = ref.opsin(lin0), ref.opsin(lin1)
    for a, b in ((x0, x1), (x0, x0)):
        om, omdc = oracle.mask(a, b)
        rm, rmdc = ref.mask(a, b)
        assert_bits_equal(om, rm, "mask")
        assert_bits_equal(omdc, rmdc, "mask_dc")


@pytest.mark.parametrize("wh", [(96, 64), (33, 41)])
def test_malta(wh):
    w, h = wh
    _, _, lin0, lin1 = linear_pair(w, h, 150, 90, qscale=10)
    p0 = ref.separate_frequencies(ref.opsin(lin0))
    p1 = ref.separate_frequencies(ref.opsin(lin1))
    cases = [(9, False, 5.1409625726 * 0.8, 5.1409625726 / 0.8, 58.5001247061),
             (7, True, 153.671655716, 153.671655716, 83150785.9592),
             (6, True, 668.358918152 * 0.9, 668.358918152 / 0.9, 0.882954368025),
             (4, True, 6841.81248144, 6841.81248144, 0.0135134962487)]
    for plane, lf, a, b, n1 in cases:
        acc = (RNG.random((h, w)) * 3).astype(np.float32)
        assert_bits_equal(oracle.malta(p0[plane], p1[plane], lf, a, b, n1, acc),
                          ref.malta(p0[plane], p1[plane], lf, a, b, n1, acc),
                          f"malta plane {plane}")


@pytest.mark.parametrize("wh,qs", [((128, 80), 4), ((64, 48), 12), ((40, 33), 2)])
def test_diffmap(wh, qs):
    w, h = wh
    _, _, lin0, lin1 = linear_pair(w, h, 60, 20, qscale=qs)
    od, os_ = oracle.diffmap(lin0, lin1)
    rd, rs = ref.diffmap(lin0, lin1)
    assert_bits_equal(od, rd, "diffmap")
    assert os_ == rs


def test_comparator_compare_and_weights():
    w, h = 136, 88
    rgb = images.crop(w, h, 220, 120)
    co = ref.encode_rgb(rgb)
    target = 0.971769
    oc, rc = oracle.comparator(rgb, target), ref.comparator(rgb, target)
    for qs in (1, 3, 9):
        q = np.full((3, 64), qs, np.int32)
        cq, _, _ = ref.reconstruct(co, w, h, q)
        od, omap = oc.compare(cq)
        rd, rmap = rc.compare(cq)
        assert_bits_equal(omap, rmap, f"distmap q={qs}")
        assert od == rd
        for direction in (1, -1):
            for r in (1, 2, 4):
                w0 = RNG.random(oc.bw * oc.bh).astype(np.float32) * (direction < 0)
                assert_bits_equal(
                    oc.block_weights(direction, r, 1.0, rmap, w0),
                    rc.block_weights(direction, r, 1.0, rmap, w0), "block weights")
    assert_bits_equal(oc.block_mask(), rc.block_mask(), "block mask")


def test_compare_block_and_zeroing_orders():
    """Phase A of SelectFrequencyMasking: CompareBlock values and the per-block candidate
    lists (processor.cc:364-467,554-590), including ragged right/bottom edge blocks."""
    w, h = 61, 43
    rgb = images.crop(w, h, 300, 150)
    target = 0.971769
    co = ref.encode_rgb(rgb)
    q = np.full((3, 64), 3, np.int32)
    cq, _, _ = ref.reconstruct(co, w, h, q)
    oc, rc = oracle.comparator(rgb, target), ref.comparator(rgb, target)
    for bx, by in [(0, 0), (3, 2), (7, 5), (7, 0), (0, 5)]:
        assert oc.compare_block(cq, bx, by) == rc.compare_block(cq, bx, by), (bx, by)
    oo, oi, oe = oc.block_zeroing_orders(cq, co)
    ro, ri, re_ = rc.block_zeroing_orders(cq, co)
    assert_bits_equal(oo, ro, "candidate offsets")
    assert_bits_equal(oi, ri, "candidate coefficient indices")
    assert_bits_equal(oe, re_, "candidate errors")
    assert ro[-1] > 100


# ------------------------------------------------------------------ YUV 4:2:0 (row f4) --
def _colourful(w, h):
    import parity_cases as pc
    return pc.colourful(w, h)


@pytest.mark.parametrize("wh", [(48, 40), (33, 35), (47, 31), (17, 16), (130, 66)])
def test_downsample_and_pixel_model_420(wh):
    """OutputImage::Downsample incl. PreProcessChannel (preprocess_downsample.cc:157-279) and the
    2x2 pixel model of UpdatePixelsForBlock (output_image.cc:146-203): the restatement keeps
    the reference's stateful pixel cache and agrees with it after scrambled update sequences
    on both sides (different ones), i.e. the cache is a function of the coefficients alone."""
    w, h = wh
    for rgb in (images.crop(w, h, 100, 60), _colourful(w, h)):
        co = ref.encode_rgb(rgb)
        exp = ref.downsample(co, w, h)
        got = oracle.downsample(co, w, h)
        assert_bits_equal(got, exp, f"downsample {w}x{h}")
        q = np.stack([RNG.integers(1, 8, 64), RNG.integers(1, 12, 64), RNG.integers(1, 12, 64)]).astype(np.int32)
        for qq in (None, q):
            for shuffle_o, shuffle_r in ((0, 0), (5, 9), (3, 0)):
                oc, osrgb, olin = oracle.reconstruct420(exp, w, h, qq, shuffle=shuffle_o)
                rc, rsrgb, rlin = ref.reconstruct420(exp, w, h, qq, shuffle=shuffle_r)
                assert_bits_equal(oc, rc, "coefficients (4:2:0)")
                assert_bits_equal(osrgb, rsrgb, "sRGB (4:2:0)")
                assert_bits_equal(olin, rlin, "linear (4:2:0)")
    # a greyscale image is left alone (output_image.cc:305-308)
    grey = np.repeat(images.crop(w, h, 100, 60)[:, :, 1:2], 3, axis=2).copy()
    cg = ref.encode_rgb(grey)
    assert_bits_equal(oracle.downsample(cg, w, h), ref.downsample(cg, w, h), "downsample (grey)")


def test_comparator_420_compare_weights_and_orders():
    w, h = 45, 27
    rgb = images.crop(w, h, 100, 60)
    co = ref.encode_rgb(rgb)
    orig = ref.downsample(co, w, h)
    cq, _, _ = ref.reconstruct420(orig, w, h, np.full((3, 64), 3, np.int32))
    oc, rc = oracle.comparator(rgb, 0.971769), ref.comparator(rgb, 0.971769)
    od, odm = oc.compare420(cq)
    rd, rdm = rc.compare420(cq)
    assert od == rd
    assert_bits_equal(odm, rdm, "distance map (4:2:0)")
    for direction in (1, -1):
        for factor in (1, 2):
            for radius in (1, 2, 4):
                assert_bits_equal(oc.block_weights_factor(direction, radius, 0.97, factor, rdm),
                                  rc.block_weights_factor(direction, radius, 0.97, factor, rdm),
                                  "block weights")
    for frame420, coeffs, og, masks in ((True, cq, orig, (1, 6)),
                                        (False, ref.reconstruct(co, w, h, np.full((3, 64), 3, np.int32))[0], co, (7, 1, 6))):
        for mask in masks:
            for lookahead, new_model in ((3, True), (2, False)):
                a = oc.block_zeroing_orders_masked(coeffs, og, frame420, mask, lookahead, new_model)
                b = rc.block_zeroing_orders_masked(coeffs, og, frame420, mask, lookahead, new_model)
                for x, y, what in zip(a, b, ("offsets", "candidates", "errors")):
                    assert_bits_equal(x, y, f"{what} frame420={frame420} mask={mask} la={lookahead}")
    oc.close()
    rc.close()
