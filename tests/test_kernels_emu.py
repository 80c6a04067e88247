"""CPU run of the product's kernel SOURCES (compiled with g++ against tests/emu/hip_emu.h)
against the oracle, through the same C ABI the GPU tests use.  Catches indexing / tiling /
accumulation-order bugs without a GPU.  This is test infrastructure, not a fallback: see
tests/emu/hip_emu.h."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import build_emu  # noqa: E402
import parity_cases as pc  # noqa: E402
from checkers import oracle, ref  # noqa: E402
from guetzli_amd.capi import Library  # noqa: E402


@pytest.fixture(scope="module")
def L():
    return Library(build_emu.build())


def test_block_kernels(L):
    pc.case_block_kernels(L, n=300)


@pytest.mark.parametrize("strips", [2, 4])
def test_reconstruct_with_several_strips_per_workgroup(L, monkeypatch, strips):
    """k_reconstruct's strip loop (large images: a workgroup takes 2 or 4 strips of 8 blocks with
    the next strip's coefficients in flight), forced on images the emulation can afford; widths
    that end inside a strip group and inside a strip."""
    monkeypatch.setenv("GZ_EMU_RECON_STRIPS", str(strips))
    pc.case_encode_quantize_reconstruct(L, 200, 43, x0=100, y0=50)
    pc.case_encode_quantize_reconstruct(L, 333, 20, x0=0, y0=50)


def test_malta_interior_and_border_tiles(L):
    """k_malta_rolled on an image with interior and border Malta tiles, against the oracle."""
    pc.case_compare(L, 200, 110, x0=100, y0=60, qscales=(5,))


def test_dct_double(L):
    pc.case_dct_double(L, n=200)


@pytest.mark.parametrize("wh", [(61, 43), (32, 32), (17, 9)])
def test_downsample_component(L, wh):
    pc.case_downsample_component(L, *wh, x0=100, y0=50)


@pytest.mark.parametrize("wh", [(61, 43), (32, 32), (72, 40)])
def test_encode_quantize_reconstruct(L, wh):
    pc.case_encode_quantize_reconstruct(L, *wh, x0=100, y0=50)


# (256, 200) and (200, 160) have tiles that take the interior (vector-load, register-window)
# path of k_blur2d for every radius; (258, 200): pitch not a multiple of 4 -> generic path
# (600, 70): tiles on the interior path of k_blur_h (radius >= 16)
@pytest.mark.parametrize("wh", [(70, 67), (300, 9), (33, 130), (256, 200), (258, 200), (600, 70)])
def test_blur(L, wh):
    pc.case_blur(L, *wh)


@pytest.mark.parametrize("wh", [(72, 48), (35, 41), (200, 160)])
def test_stages(L, wh):
    pc.case_stages(L, *wh)


def test_compare(L):
    pc.case_compare(L, 80, 56, x0=200, y0=100, qscales=(1, 5))


def test_config_struct_on_one_context(L):
    """gz_get_config / gz_set_config (include/guetzli_amd.h, round 6): the instantiation switches flipped on ONE
    context between evaluations -- same distance, map and block maxima bit for bit; bad values are refused."""
    import images
    import numpy as np
    rgb = images.crop(96, 56, 200, 100)
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb(download=False)
        ctx.quantize(np.full((3, 64), 5, np.int32), download=False)
        base = ctx.get_config().as_dict()
        d0, dm0, bm0 = ctx.compare()
        for kw in (dict(blur_packed=1, tile_rows=32), dict(blur_packed=0, tile_rows=16, single_stream=1), dict(single_stream=0),
                   dict(side_small=1, store_distmap=1)):
            ctx.set_config(**dict(base, **kw))
            assert ctx.get_config().as_dict() == dict(base, **kw)
            d, dm, bm = ctx.compare()
            assert d == d0
            pc.assert_bits_equal(dm, dm0, f"distance map under {kw}")
            pc.assert_bits_equal(bm, bm0, f"block maxima under {kw}")
            ctx.compare_begin()
            assert ctx.compare_end() == d0
        with pytest.raises(Exception):
            ctx.set_config(tile_rows=24)
        with pytest.raises(Exception):
            ctx.set_config(struct_size=4)


def test_stages_and_compare_with_32_row_tiles(L, monkeypatch):
    """Images this small take the 16-row blur tiles by default; the 32-row instantiations (what
    1080p and 4K run) are forced here so that both are checked in emulation."""
    monkeypatch.setenv("GZ_TILE_ROWS", "32")
    pc.case_stages(L, 72, 48)
    pc.case_compare(L, 80, 56, x0=200, y0=100, qscales=(5,))


def test_block_search(L):
    pc.case_block_search(L, 45, 27)


def test_compare_blocks(L):
    pc.case_compare_blocks(L, 45, 27, x0=100, y0=60)


@pytest.fixture(scope="module")
def host_emu():
    from guetzli_amd.encoder import HostLibrary
    return HostLibrary(build_emu.build_host())


@pytest.mark.parametrize("wh", [(61, 43), (32, 32), (129, 9), (8, 8), (448, 296)])   # the last: 2072 MCUs = two scan tiles
def test_jpeg_entropy(L, host_emu, wh):
    pc.case_jpeg_entropy(L, host_emu, *wh, x0=100, y0=50)


def test_jpeg_histograms_generic_kernel(L, host_emu, monkeypatch):
    """k_jpeg_histograms (the frame's geometry interpreted per block, an integer division per lane) stays
    in the library as the reference of k_jpeg_histograms_t<blocks per MCU>, which the chain uses."""
    monkeypatch.setenv("GZ_HIST_GENERIC", "1")
    pc.case_jpeg_entropy(L, host_emu, 61, 43, x0=100, y0=50)
    if ref is not None:
        pc.case_jpeg_entropy420(L, host_emu, 48, 40, ref, x0=100, y0=50)


def test_global_order(L):
    pc.case_global_order(L, 40, 32, x0=100, y0=60)


def test_blur_and_compare_with_paired_row_column_passes(L, monkeypatch):
    """The row-pair / column-pair passes with packed arithmetic (k_blur_h_pk, k_blur_v_pk: what
    images from 4 MPix on run) forced on images small enough for the emulation but wide and
    tall enough to have interior tiles (x0 >= 256, 16-row tiles away from the border rows)."""
    monkeypatch.setenv("GZ_BLUR_PK", "1")
    pc.case_blur(L, 840, 72)
    monkeypatch.setenv("GZ_TILE_ROWS", "32")
    pc.case_blur(L, 600, 100, configs=pc.SIGMAS_BR[:4])
    pc.case_stages(L, 72, 48)
    pc.case_compare(L, 80, 56, x0=200, y0=100, qscales=(5,))


# ------------------------------------------------------------------ YUV 4:2:0 (row f4) --
needs_ref = pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")


@pytest.mark.parametrize("wh", [(48, 40), (33, 35), (47, 31), (64, 48)])
def test_frame420(L, wh):
    pc.case_frame420(L, *wh, oracle, x0=100, y0=60)


def test_frame420_preprocessing_branches(L):
    """An image on which PreProcessChannel sharpens and blurs (not just passes through)."""
    rgb = pc.colourful(72, 56)
    co = oracle.encode_rgb(rgb)
    plain_u, plain_v = oracle.downsample_chroma(co, 72, 56, 2, 2)
    full = oracle.downsample(co, 72, 56)
    nb = 9 * 7
    assert (full[nb:nb + 20] != plain_u).any() and (full[nb + 20:] != plain_v).any(), \
        "the test image does not exercise the pre-processing"
    pc.case_frame420(L, 72, 56, oracle, rgb=rgb)


def test_block_search420(L):
    pc.case_block_search420(L, 45, 27, oracle, x0=100, y0=60)
    pc.case_block_search420(L, 40, 33, oracle, x0=10, y0=10, qs=2, lookahead=2, new_model=False)


def test_block_search_masks_and_params(L):
    pc.case_block_search_masks444(L, 40, 24, oracle, x0=100, y0=60)
    pc.case_block_search_masks444(L, 33, 17, oracle, x0=50, y0=60, lookahead=1)
    pc.case_block_search_masks444(L, 33, 17, oracle, x0=50, y0=60, lookahead=5, new_model=False)


@needs_ref
@pytest.mark.parametrize("wh", [(61, 43), (32, 32), (48, 40), (129, 9)])
def test_jpeg_entropy420(L, host_emu, wh):
    pc.case_jpeg_entropy420(L, host_emu, *wh, ref, x0=100, y0=50)


@pytest.mark.parametrize("wh", [(40, 32), (45, 35)])
def test_patched_candidate_planes(L, wh):
    pc.case_patched_candidate_planes(L, *wh, x0=100, y0=60)


def test_global_order420(L):
    pc.case_global_order420(L, 48, 40, oracle, x0=100, y0=60)
