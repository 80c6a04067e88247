"""guetzli_amd/host/jpeg_reader.cc against guetzli::ReadJpeg (the unmodified reference behind
oracle/_ref): every field Process(jpeg_data) consumes -- dimensions, sampling, quantisation
tables, quantised coefficients, APPn / COM / tail bytes -- on JPEGs written by Pillow
(baseline, progressive, optimised tables, restart intervals, 4:4:4 / 4:2:2 / 4:2:0, grey,
odd sizes, metadata).  CPU only."""
import ctypes as C
import io
import os

import numpy as np
import pytest
from PIL import Image

import images
from checkers import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")


@pytest.fixture(scope="module")
def host():
    from guetzli_amd import build as gzbuild
    lib = C.CDLL(gzbuild.build_host())
    lib.gzh_read_jpeg.restype = C.c_long
    lib.gzh_read_jpeg.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    return lib


def dump(fn, data):
    buf = np.frombuffer(data, np.uint8)
    cap = 1 << 26
    out = np.zeros(cap, np.uint8)
    n = fn(buf.ctypes.data, len(data), out.ctypes.data, cap)
    return None if n < 0 else out[:n].tobytes()


def ref_dump(data):
    f = ref.lib.ref_read_jpeg
    f.restype, f.argtypes = C.c_long, [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    return dump(f, data)


def jpeg_bytes(rgb, **kw):
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **kw)
    return b.getvalue()


CASES = [
    dict(quality=90, subsampling=0),
    dict(quality=75, subsampling=0, optimize=True),
    dict(quality=95, subsampling=0, progressive=True),
    dict(quality=60, subsampling=0, progressive=True, optimize=True),
    dict(quality=85, subsampling=2),                       # 4:2:0
    dict(quality=85, subsampling=1, progressive=True),     # 4:2:2
    dict(quality=30, subsampling=0),
    dict(quality=100, subsampling=0),
    dict(quality=92, subsampling=0, restart_marker_blocks=7),
    dict(quality=92, subsampling=2, progressive=True, restart_marker_rows=1),
    dict(quality=88, subsampling=0, comment=b"a comment", dpi=(72, 300)),
]


@pytest.mark.parametrize("kw", CASES, ids=[str(i) for i in range(len(CASES))])
@pytest.mark.parametrize("wh", [(96, 64), (61, 43), (17, 9), (8, 8)])
def test_reader_matches_reference(host, kw, wh):
    w, h = wh
    rgb = images.crop(w, h, 120, 80)
    try:
        data = jpeg_bytes(rgb, **kw)
    except TypeError:
        pytest.skip("this Pillow does not know an option of the case")
    exp = ref_dump(data)
    got = dump(host.gzh_read_jpeg, data)
    assert got == exp   # None on both sides when the reference rejects the stream


def test_grey_exif_and_tail(host):
    rgb = images.crop(80, 56, 10, 10)
    grey = jpeg_bytes(np.ascontiguousarray(rgb[:, :, 1]), quality=90)
    assert dump(host.gzh_read_jpeg, grey) == ref_dump(grey)
    exif = b"Exif\x00\x00" + bytes(range(200))
    data = jpeg_bytes(rgb, quality=90, subsampling=0, exif=exif) + b"trailing bytes after EOI"
    exp = ref_dump(data)
    assert exp is not None and dump(host.gzh_read_jpeg, data) == exp


def test_malformed_streams_are_rejected_like_the_reference(host):
    rgb = images.crop(64, 48, 30, 30)
    good = jpeg_bytes(rgb, quality=90, subsampling=0)
    for bad in (good[:2], good[: len(good) // 2], b"\x00" + good, good[:-2],
                good.replace(b"\xff\xc0", b"\xff\xc3", 1)):
        assert (dump(host.gzh_read_jpeg, bad) is None) == (ref_dump(bad) is None)


def test_divergences_the_fuzzer_found_stay_fixed(host):
    """Hand-made from a baseline stream, after tests/test_fuzz_readers.py's campaigns: (1) a marker the reference does
    not know (0xff 0xf2 in place of APP0) is skipped like garbage; (2) a DC Huffman table that is never defined while
    the scan header names a band without DC (Ss = 4): the header check passes -- it goes by the header's band even in
    a sequential frame -- and the first DC symbol then meets an empty table: refused; (3) the same symbol twice in a
    DHT: refused; (4) a fifth quantisation table: refused."""
    rgb = images.crop(48, 40, 60, 60)
    good = jpeg_bytes(rgb, quality=90, subsampling=0)
    assert ref_dump(good) is not None

    def verdicts(data):
        exp, got = ref_dump(data), dump(host.gzh_read_jpeg, data)
        assert got == exp
        return exp

    # (1)
    i = good.index(b"\xff\xe0")
    assert verdicts(good[:i + 1] + b"\xf2" + good[i + 2:]) is not None
    # (2)
    d0 = good.index(b"\xff\xc4")                       # the first DHT: DC table 0
    assert good[d0 + 4] == 0x00
    sos = good.index(b"\xff\xda")
    n = good[sos + 4]
    hdr = sos + 5 + 2 * n                               # Ss, Se, AhAl
    bad = bytearray(good)
    bad[d0 + 1] = 0x5c
    bad[hdr], bad[hdr + 1] = 4, 5
    assert verdicts(bytes(bad)) is None
    # (3)
    bad = bytearray(good)
    bad[d0 + 5 + 16 + 1] = bad[d0 + 5 + 16]             # second symbol := first symbol
    assert verdicts(bytes(bad)) is None
    # (4)
    q0 = good.index(b"\xff\xdb")
    qlen = (good[q0 + 2] << 8) | good[q0 + 3]
    seg = good[q0:q0 + 2 + qlen]
    many = good[:q0] + seg * 5 + good[q0 + 2 + qlen:]
    assert verdicts(many) is None
