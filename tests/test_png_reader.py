"""guetzli_amd/host/png_reader.cc = ReadPNG of the reference's front end (guetzli.cc:47-152:
libpng's png_read_png with PACKING | EXPAND | STRIP_16, then alpha blended on black).

Pinned against the reference function itself: oracle/_ref/libgz_ref_png.so is guetzli.cc's
ReadPNG compiled where it lies, over the image's libpng 1.6.37 (oracle/ref_png_harness.cc) --
every file of this suite goes through both, pixels and accept / reject decisions must agree.
Beside that, independent of libpng: (a) Pillow's decoder on files written by Pillow and on
hand-assembled files of every colour type / bit depth / interlace / filter combination, with
the reference's post-processing (alpha on black, 16 -> 8 by the high byte) applied to
Pillow's samples, and (b) the raw samples the hand-assembled files were made from.  CPU only."""
import ctypes as C
import io
import os
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

import images

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    from guetzli_amd import build as gzbuild
    from guetzli_amd.encoder import HostLibrary
    return CheckedHost(HostLibrary(gzbuild.build_host()))


def test_the_reference_reader_is_available():
    """Where the oracle was built with libpng, say so (the pin of this suite); elsewhere the
    suite still runs against Pillow and the raw samples."""
    if REF_PNG is None:
        pytest.skip("oracle/_ref/libgz_ref_png.so not built: libpng-independent checks only")
    assert via_reference(open(images.BEES, "rb").read()).shape == (258, 444, 3)


REF_PNG = None
_p = os.path.join(ROOT, "oracle", "_ref", "libgz_ref_png.so")
if os.path.exists(_p):
    try:
        REF_PNG = C.CDLL(_p)
        REF_PNG.ref_read_png.restype = C.c_long
        REF_PNG.ref_read_png.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long]
    except OSError:
        REF_PNG = None


def via_reference(data):
    """guetzli.cc's ReadPNG (libpng): uint8 [h][w][3], or None if it refuses the stream."""
    buf = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
    wh = (C.c_int * 2)()
    n = REF_PNG.ref_read_png(buf.ctypes.data, len(data), wh, None, 0)
    if n < 0:
        return None
    out = np.zeros(n, np.uint8)
    assert REF_PNG.ref_read_png(buf.ctypes.data, len(data), wh, out.ctypes.data, n) == n
    return out.reshape(wh[1], wh[0], 3)


class CheckedHost:
    """The product's reader, with every call also checked against the reference's."""

    def __init__(self, host):
        self.host = host

    def read_png(self, data):
        try:
            got = self.host.read_png(data)
        except ValueError:
            if REF_PNG is not None:
                assert via_reference(data) is None, "the reference accepts what the product refuses"
            raise
        if REF_PNG is not None:
            exp = via_reference(data)
            assert exp is not None, "the reference refuses what the product accepts"
            assert exp.shape == got.shape and np.array_equal(exp, got), "pixels differ from the reference's ReadPNG"
        return got


def blend(v, a):   # BlendOnBlack, guetzli.cc:42-44
    return ((v.astype(np.int32) * a.astype(np.int32) + 128) // 255).astype(np.uint8)


def via_pillow(data):
    """Pillow's decode + the reference's channel handling."""
    im = Image.open(io.BytesIO(data))
    im.load()
    if im.mode in ("I;16", "I;16B", "I"):
        g = (np.array(im).astype(np.uint32) >> 8).astype(np.uint8)
        return np.stack([g, g, g], -1)
    has_alpha = im.mode in ("LA", "RGBA", "PA") or "transparency" in im.info
    a = np.array(im.convert("RGBA"))
    if not has_alpha:
        return np.ascontiguousarray(a[..., :3])
    return np.stack([blend(a[..., c], a[..., 3]) for c in range(3)], -1)


# ---- a minimal PNG writer: any colour type / depth / filter / interlace ----------------
def chunk(kind, body, bad_crc=False):
    crc = zlib.crc32(kind + body) & 0xffffffff
    if bad_crc:
        crc ^= 0x5a5a5a5a
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", crc)


def pack_row(samples, depth):
    """samples: 1-D array of per-sample values in file order."""
    if depth == 8:
        return samples.astype(np.uint8).tobytes()
    if depth == 16:
        return samples.astype(">u2").tobytes()
    per = 8 // depth
    pad = (-len(samples)) % per
    s = np.concatenate([samples, np.zeros(pad, samples.dtype)]).astype(np.uint32).reshape(-1, per)
    out = np.zeros(len(s), np.uint32)
    for k in range(per):
        out |= s[:, k] << (depth * (per - 1 - k))
    return out.astype(np.uint8).tobytes()


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_row(ftype, row, prev, bpp):
    row = bytearray(row)
    out = bytearray(len(row))
    for i in range(len(row)):
        a = row[i - bpp] if i >= bpp else 0
        b = prev[i]
        c = prev[i - bpp] if i >= bpp else 0
        pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ftype] if ftype < 5 else 0   # > 4: invalid on purpose
        out[i] = (row[i] - pred) & 0xff
    return bytes(out)


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def make_png(samples, color_type, depth, palette=None, trns=None, interlace=False, filters=(0,),
             extra_chunks=(), iend=True, idat_split=1, bad_idat_crc=False):
    """samples: [h][w][channels] integer array of raw sample values."""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    raw = bytearray()
    passes = ADAM7 if interlace else [(0, 0, 1, 1)]
    k = 0
    for (x0, y0, dx, dy) in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = bytes(len(pack_row(sub[0].reshape(-1), depth)))
        for r in range(sub.shape[0]):
            row = pack_row(sub[r].reshape(-1), depth)
            f = filters[k % len(filters)]
            k += 1
            raw += bytes([f]) + filter_row(f, row, prev, bpp)
            prev = row
    comp = zlib.compress(bytes(raw), 6)
    out = b"\x89PNG\r\n\x1a\n"
    out += chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    for c in extra_chunks:
        out += c
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", trns)
    n = len(comp)
    cuts = [n * i // idat_split for i in range(idat_split + 1)]
    for i in range(idat_split):
        out += chunk(b"IDAT", comp[cuts[i]:cuts[i + 1]], bad_crc=bad_idat_crc and i == 0)
    if iend:
        out += chunk(b"IEND", b"")
    return out


def expected_from_samples(samples, color_type, depth, palette=None, trns=None):
    """libpng's PACKING | EXPAND | STRIP_16 on the raw samples, then ReadPNG's switch."""
    s = samples.astype(np.uint32)
    h, w, _ = s.shape
    alpha = None
    if color_type == 3:
        pal = np.zeros((256, 3), np.uint8)
        pal[:len(palette)] = np.asarray(palette, np.uint8).reshape(-1, 3)
        rgbv = pal[s[..., 0]]
        if trns is not None:
            al = np.full(256, 255, np.uint8)
            al[:len(trns)] = np.frombuffer(trns, np.uint8)
            alpha = al[s[..., 0]]
    elif color_type in (0, 4):
        v = s[..., 0]
        scale = {1: 0xff, 2: 0x55, 4: 0x11, 8: 1, 16: 1}[depth]
        v8 = (v >> 8 if depth == 16 else v * scale).astype(np.uint8)
        rgbv = np.stack([v8, v8, v8], -1)
        if color_type == 4:
            alpha = (s[..., 1] >> 8 if depth == 16 else s[..., 1]).astype(np.uint8)
        elif trns is not None:
            key = struct.unpack(">H", trns)[0]
            if depth < 16:
                key = ((key & ((1 << depth) - 1)) * scale) & 0xff
                alpha = np.where(v8 == key, 0, 255).astype(np.uint8)
            else:
                alpha = np.where(v == key, 0, 255).astype(np.uint8)
    else:
        v = s[..., :3]
        rgbv = (v >> 8 if depth == 16 else v).astype(np.uint8)
        if color_type == 6:
            alpha = (s[..., 3] >> 8 if depth == 16 else s[..., 3]).astype(np.uint8)
        elif trns is not None:
            key = np.array(struct.unpack(">HHH", trns), np.uint32)
            if depth == 8:
                key &= 0xff
            alpha = np.where((v == key).all(-1), 0, 255).astype(np.uint8)
    if alpha is None:
        return np.ascontiguousarray(rgbv)
    return np.stack([blend(rgbv[..., c], alpha) for c in range(3)], -1)


def rnd(shape, hi, seed):
    return np.random.RandomState(seed).randint(0, hi, size=shape).astype(np.uint32)


# ---- files written by Pillow ------------------------------------------------------------
def pillow_png(im, **kw):
    b = io.BytesIO()
    im.save(b, "PNG", **kw)
    return b.getvalue()


def test_bees_png(host):
    data = open(images.BEES, "rb").read()
    got = host.read_png(data)
    assert got.shape == (258, 444, 3)
    assert np.array_equal(got, images.bees())


PILLOW_CASES = {
    "rgb": lambda: Image.fromarray(images.crop(61, 43, 100, 50)),
    "rgba": lambda: Image.fromarray(np.dstack([images.crop(40, 33, 10, 20),
                                              rnd((33, 40), 256, 1).astype(np.uint8)])),
    "grey": lambda: Image.fromarray(images.crop(37, 29)[..., 1]),
    "grey_alpha": lambda: Image.fromarray(np.dstack([images.crop(37, 29)[..., 0],
                                                    rnd((29, 37), 256, 2).astype(np.uint8)]), "LA"),
    "bilevel": lambda: Image.fromarray(images.crop(67, 21)[..., 1] > 120),
    "palette": lambda: Image.fromarray(images.crop(50, 50, 30, 30)).quantize(200),
    "palette16": lambda: Image.fromarray(images.crop(51, 17, 30, 30)).quantize(13),
    "palette4": lambda: Image.fromarray(images.crop(33, 9, 30, 30)).quantize(4),
    "palette2": lambda: Image.fromarray(images.crop(35, 7, 30, 30)).quantize(2),
}


@pytest.mark.parametrize("name", sorted(PILLOW_CASES))
def test_files_written_by_pillow(host, name):
    data = pillow_png(PILLOW_CASES[name](), optimize=name in ("rgb", "palette16"))
    assert np.array_equal(host.read_png(data), via_pillow(data))


def test_palette_with_transparency_written_by_pillow(host):
    im = Image.fromarray(images.crop(48, 40, 60, 60)).quantize(64)
    data = pillow_png(im, transparency=bytes(rnd((40,), 256, 3).astype(np.uint8)))
    assert np.array_equal(host.read_png(data), via_pillow(data))


# ---- hand-assembled files: every colour type / depth, filters, interlace ------------------
COMBOS = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8),
          (4, 8), (4, 16), (6, 8), (6, 16)]
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("combo", COMBOS)
def test_every_colour_type_and_depth(host, combo, interlace):
    ct, depth = combo
    for (w, h) in [(1, 1), (5, 3), (19, 11), (33, 8)]:
        s = rnd((h, w, CHANNELS[ct]), 1 << depth, 7 * w + h + depth)
        palette = rnd((min(1 << depth, 200), 3), 256, 9) if ct == 3 else None
        if ct == 3:
            s %= len(palette)
        data = make_png(s, ct, depth, palette=palette, interlace=interlace, filters=(0, 1, 2, 3, 4),
                        idat_split=3)
        got = host.read_png(data)
        exp = expected_from_samples(s, ct, depth, palette=palette)
        assert got.shape == exp.shape and np.array_equal(got, exp), (combo, interlace, w, h)
        assert np.array_equal(got, via_pillow(data)), ("pillow", combo, interlace, w, h)


@pytest.mark.parametrize("combo", [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 4), (3, 8)])
def test_trns_becomes_alpha(host, combo):
    ct, depth = combo
    w, h = 23, 9
    s = rnd((h, w, CHANNELS[ct]), min(1 << depth, 4), 11 + depth)   # few values: the key is hit
    palette = trns = None
    if ct == 3:
        palette = rnd((4, 3), 256, 5)
        trns = bytes([0, 128, 255])            # shorter than the palette: the rest is opaque
    elif ct == 0:
        trns = struct.pack(">H", 1)
    else:
        trns = struct.pack(">HHH", 1, 2, 3)
        s[0, 0] = (1, 2, 3)
        s[2, 5] = (1, 2, 3)
    data = make_png(s, ct, depth, palette=palette, trns=trns, filters=(4, 2))
    got = host.read_png(data)
    assert np.array_equal(got, expected_from_samples(s, ct, depth, palette=palette, trns=trns))
    # Pillow keeps the grey key unscaled next to samples it has scaled to 8 bits, so below 8
    # bits it never matches, and it drops the key of 16-bit grey; libpng scales the key with
    # the samples (png_do_expand)
    if not (ct == 0 and depth != 8):
        assert np.array_equal(got, via_pillow(data))


def test_sixteen_bit_keys_are_compared_in_full(host):
    s = np.array([[[0x1234], [0x12ff], [0x1234]]], np.uint32)
    data = make_png(s, 0, 16, trns=struct.pack(">H", 0x1234))
    exp = np.array([[[0, 0, 0], [0x12] * 3, [0, 0, 0]]], np.uint8)
    assert np.array_equal(host.read_png(data), exp)


def test_palette_index_beyond_the_palette_is_black(host):
    s = np.array([[[0], [1], [7]]], np.uint32)
    data = make_png(s, 3, 4, palette=[[10, 20, 30], [40, 50, 60]])
    assert np.array_equal(host.read_png(data), np.array([[[10, 20, 30], [40, 50, 60], [0, 0, 0]]], np.uint8))


def test_ancillary_chunks_are_skipped_and_a_bad_ancillary_crc_is_tolerated(host):
    s = rnd((6, 7, 3), 256, 21)
    extra = [chunk(b"gAMA", struct.pack(">I", 45455)), chunk(b"tEXt", b"Comment\0hello"),
             chunk(b"pHYs", struct.pack(">IIB", 1, 1, 0), bad_crc=True)]
    data = make_png(s, 2, 8, extra_chunks=extra) + b"trailing bytes"
    assert np.array_equal(host.read_png(data), s.astype(np.uint8))


def _rejected(host, data):
    try:
        host.read_png(data)
    except ValueError:
        return True
    return False


def test_refusals(host, capfd):
    s = rnd((4, 4, 3), 256, 22)
    good = make_png(s, 2, 8)
    assert not _rejected(host, good)
    assert _rejected(host, b"")
    assert _rejected(host, b"\xff\xd8\xff\xe0 not a png")
    assert _rejected(host, good[:40])                                   # truncated
    assert _rejected(host, make_png(s, 2, 8, iend=False))               # no IEND
    assert _rejected(host, make_png(s, 2, 8, bad_idat_crc=True))        # critical CRC
    assert _rejected(host, make_png(s, 2, 8, filters=(5,)))             # bad filter byte
    assert _rejected(host, make_png(rnd((4, 4, 1), 4, 1), 3, 2))        # palette image without PLTE
    assert _rejected(host, make_png(s, 2, 8, extra_chunks=[chunk(b"ABCD", b"x")]))   # unknown critical chunk
    bad_depth = bytearray(good)
    bad_depth[24] = 4                                                   # RGB with 4 bits
    bad_depth[29:33] = struct.pack(">I", zlib.crc32(bytes(bad_depth[12:29])) & 0xffffffff)
    assert _rejected(host, bytes(bad_depth))
    zero_w = bytearray(good)
    zero_w[16:20] = struct.pack(">I", 0)
    zero_w[29:33] = struct.pack(">I", zlib.crc32(bytes(zero_w[12:29])) & 0xffffffff)
    assert _rejected(host, bytes(zero_w))
    # image data shorter than the header promises
    short = make_png(s[:2], 2, 8)
    tall = bytearray(short)
    tall[20:24] = struct.pack(">I", 4)
    tall[29:33] = struct.pack(">I", zlib.crc32(bytes(tall[12:29])) & 0xffffffff)
    assert _rejected(host, bytes(tall))
    capfd.readouterr()


def test_chunk_order_and_oddities_follow_libpng(host):
    """Streams on the edges of the specification: whatever the product does, CheckedHost holds
    it to what the reference's libpng does (accept with the same pixels, or refuse)."""
    s = rnd((5, 6, 1), 4, 31)
    pal = rnd((4, 3), 256, 32)
    plte = chunk(b"PLTE", np.asarray(pal, np.uint8).tobytes())
    trns = chunk(b"tRNS", bytes([0, 100]))
    ihdr = chunk(b"IHDR", struct.pack(">IIBBBBB", 6, 5, 2, 3, 0, 0, 0))
    raw = b"".join(bytes([0]) + pack_row(s[r].reshape(-1), 2) for r in range(5))
    idat = chunk(b"IDAT", zlib.compress(raw))
    iend = chunk(b"IEND", b"")
    sig = b"\x89PNG\r\n\x1a\n"
    cases = {
        "plain": sig + ihdr + plte + trns + idat + iend,
        "tRNS before PLTE": sig + ihdr + trns + plte + idat + iend,
        "tRNS after IDAT": sig + ihdr + plte + idat + trns + iend,
        "duplicate tRNS": sig + ihdr + plte + trns + chunk(b"tRNS", bytes([255, 255, 0])) + idat + iend,
        "duplicate PLTE": sig + ihdr + plte + plte + idat + iend,
        "PLTE after IDAT": sig + ihdr + idat + plte + iend,
        "empty IDAT chunks": sig + ihdr + plte + chunk(b"IDAT", b"") + idat + chunk(b"IDAT", b"") + iend,
        "too much image data": sig + ihdr + plte + chunk(b"IDAT", zlib.compress(raw + b"\0" * 40)) + iend,
        "palette longer than the depth allows": sig + ihdr + chunk(b"PLTE", bytes(range(30))) + idat + iend,
        "IHDR twice": sig + ihdr + ihdr + plte + idat + iend,
        "no IHDR": sig + plte + idat + iend,
        "bad IHDR length": sig + chunk(b"IHDR", struct.pack(">IIBBBBBB", 6, 5, 2, 3, 0, 0, 0, 0)) + plte + idat + iend,
        "interlace method 2": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 6, 5, 2, 3, 0, 0, 2)) + plte + idat + iend,
        "width above the user limit": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 1000001, 1, 8, 0, 0, 0, 0)) +
                                      chunk(b"IDAT", zlib.compress(bytes(1000002))) + iend,
    }
    grey = rnd((4, 7, 1), 256, 33)
    cases["PLTE in a greyscale image"] = make_png(grey, 0, 8, extra_chunks=[plte])
    cases["tRNS with an alpha channel"] = make_png(rnd((3, 3, 2), 256, 34), 4, 8, trns=struct.pack(">H", 5))
    cases["tRNS of the wrong length"] = make_png(grey, 0, 8, trns=b"\0\1\2")
    cases["bKGD and friends"] = make_png(grey, 0, 8, extra_chunks=[chunk(b"bKGD", struct.pack(">H", 7)),
                                                                   chunk(b"sBIT", b"\x05"),
                                                                   chunk(b"tIME", bytes(7))])
    if REF_PNG is None:
        pytest.skip("needs the reference's ReadPNG as the judge of these streams")
    for name, data in cases.items():
        try:
            host.read_png(data)
        except ValueError:
            pass
        except AssertionError as e:
            raise AssertionError(f"{name}: {e}")


def _rechunk(data, mutate):
    """Applies mutate(type, body) -> body to every chunk and recomputes the CRCs."""
    out = data[:8]
    pos = 8
    while pos + 12 <= len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        kind = data[pos + 4:pos + 8]
        body = mutate(kind, data[pos + 8:pos + 8 + n])
        out += chunk(kind, body)
        pos += 12 + n
    return out


def test_corrupted_streams_get_the_reference_verdict(host):
    """300 single-byte corruptions inside chunk bodies (CRCs recomputed, so that the damage
    reaches the parser, the inflater and the filters): accepted with the reference's pixels, or
    refused like the reference does."""
    if REF_PNG is None:
        pytest.skip("needs the reference's ReadPNG as the judge")
    rs = np.random.RandomState(77)
    seeds = [make_png(rnd((9, 11, 3), 256, 41), 2, 8, filters=(0, 1, 2, 3, 4)),
             make_png(rnd((7, 13, 1), 16, 42) % 9, 3, 4, palette=rnd((9, 3), 256, 43), trns=bytes([3, 200]),
                      interlace=True, filters=(4, 3)),
             make_png(rnd((6, 5, 2), 65536, 44), 4, 16, interlace=True, idat_split=2)]
    verdicts = {"accepted": 0, "refused": 0}
    for k in range(300):
        base = seeds[k % len(seeds)]
        target = rs.randint(0, 1 << 30)
        state = {"i": 0}

        def mutate(kind, body):
            if not body or kind == b"IEND":
                return body
            state["i"] += 1
            if (target + state["i"]) % 3 != 0:
                return body
            b = bytearray(body)
            j = (target // 7) % len(b)
            b[j] ^= 1 << ((target // 3) % 8)
            return bytes(b)

        data = _rechunk(base, mutate)
        try:
            host.read_png(data)            # CheckedHost compares with the reference
            verdicts["accepted"] += 1
        except ValueError:
            verdicts["refused"] += 1
    assert verdicts["accepted"] > 20 and verdicts["refused"] > 20, verdicts
