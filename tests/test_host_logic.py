"""Host search-driver building blocks that must reproduce libstdc++ / reference behaviour
exactly (CPU only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("keys", ["float_second_key", "plain"])
def test_lazy_sort_is_std_sort(tmp_path, keys):
    """guetzli_amd/host/lazy_sort.h yields std::sort's permutation (ties included): the
    C++ check in tests/cpp/test_lazy_sort.cc compares against std::sort itself -- with the
    comparator tagged as phase B's is (the partitions' pass over the keys runs eight entries at a
    time with AVX2) and untagged (the generic pass)."""
    exe = str(tmp_path / "test_lazy_sort")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread"] +
                   (["-DGZ_TEST_PLAIN_LESS"] if keys == "plain" else []) +
                   [os.path.join(ROOT, "tests", "cpp", "test_lazy_sort.cc"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "lazy_sort: ok" in out.stdout


def test_local_ac_symbol_update_equals_two_block_passes(tmp_path):
    """ReplaceCoeffACSymbols (the symbols around one changed coefficient, phase B's slow steps)
    == AddBlockACSymbols(-1) + store + AddBlockACSymbols(+1) on 200 000 random blocks."""
    exe = str(tmp_path / "test_ac_symbols")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall",
                    os.path.join(ROOT, "tests", "cpp", "test_ac_symbols.cc"),
                    os.path.join(ROOT, "guetzli_amd", "host", "jpeg_writer.cc"), "-o", exe, "-lz"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ac_symbols: ok" in out.stdout


def test_code_refresh_hand_over_under_thread_sanitizer(tmp_path):
    """CodeRefreshers (guetzli_amd/host/code_refresh.h): 4 852 code refreshes through 1..4 helper threads,
    several windows in flight, the helpers put to sleep and woken between bursts -- every result equal
    to EntropyCodes / HistogramRawBits on the spot, and no data race (-fsanitize=thread)."""
    exe = str(tmp_path / "test_code_refresh")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-pthread",
           os.path.join(ROOT, "tests", "cpp", "test_code_refresh.cc"),
           os.path.join(ROOT, "guetzli_amd", "host", "jpeg_writer.cc"), "-o", exe, "-lz"]
    tsan = subprocess.run(cmd + ["-fsanitize=thread"], capture_output=True, text=True)
    if tsan.returncode != 0:   # (a toolchain without the sanitizer's runtime: the functional check alone)
        subprocess.run(cmd, check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "code_refresh: ok" in out.stdout
    assert "ThreadSanitizer" not in out.stderr


def test_device_partition_is_std_sort_in_emulation(tmp_path):
    """gz_order_partition (gz_kernels_order.h, here the CPU emulation build of the kernel
    sources) driven by LazySorted reproduces std::sort's permutation, ties included."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    exe = str(tmp_path / "test_device_order")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread",
                    os.path.join(ROOT, "tests", "cpp", "test_device_order.cc"), "-o", exe, "-ldl"],
                   check=True)
    out = subprocess.run([exe, build_emu.build(), "32769", "512"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device_order: ok" in out.stdout
    # k_desc_swap's instantiation with the larger tables (on the GPU: orders beyond 8.4 M entries)
    out = subprocess.run([exe, build_emu.build(), "20000", "512"], capture_output=True, text=True,
                         env=dict(os.environ, GZ_EMU_DESC_BIG="1", GZ_TEST_ONLY_N="20000"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device_order: ok" in out.stdout
    # k_desc_swap's loop over groups of pairs (on the GPU: orders beyond 2 M entries): a grid of 3
    out = subprocess.run([exe, build_emu.build(), "20000", "512"], capture_output=True, text=True,
                         env=dict(os.environ, GZ_EMU_DESC_SWAP_GRID="3", GZ_TEST_ONLY_N="20000"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device_order: ok" in out.stdout


def _build_rank_sort(tmp_path):
    exe = str(tmp_path / "test_rank_sort")
    subprocess.run(["g++", "-O2", "-std=c++17", "-DGZ_EMU", "-I" + os.path.join(ROOT, "tests", "emu"),
                    "-I" + os.path.join(ROOT, "guetzli_amd", "csrc"), "-pthread",
                    os.path.join(ROOT, "tests", "cpp", "test_rank_sort.cc"), "-o", exe, "-ldl"],
                   check=True)
    return exe


def test_device_ranking_is_std_sort_in_emulation(tmp_path):
    """gz_kernels_rank.h (the per-block std::sort of the zeroing candidates, heap-sort
    fall-back included) against std::sort itself: the restatement compiled for the host, and
    the emulation build's kernel through gz_probe_rank_sort."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    out = subprocess.run([_build_rank_sort(tmp_path), build_emu.build()], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "restatement == std::sort" in out.stdout and "device == std::sort" in out.stdout


def test_opsin_division_identities_hold_for_every_float(tmp_path):
    """gz_math.h evaluates two of the reference's three FP64 divisions per channel of the opsin
    sensitivity by cheaper sequences (a division by a constant through two FMAs; a double
    quotient of floats rounded to float as the float quotient).  tests/cpp/
    verify_opsin_divisions.cc checks the first on ALL 2^32 float inputs and the second on 10^9
    random / in-range pairs against the plain C++ divisions (x86-64, as the reference runs)."""
    exe = str(tmp_path / "verify_div")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-pthread",
                    "-I" + os.path.join(ROOT, "guetzli_amd", "csrc"), "-I" + os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "cpp", "verify_opsin_divisions.cc"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(" 0 mismatches") == 3, out.stdout   # the third: gamma_poly_f as a whole, every float


def test_malta_diff_fast_form_equals_the_reference_sequence(tmp_path):
    """gz_math.h evaluates Malta's per-pixel "diffs" value (MaltaDiffMapImpl, butteraugli.cc:
    1473-1529) without the reference's if-ladder, with a float form of `absval` and with both
    quotients from one reciprocal.  tests/cpp/verify_malta_diff.cc checks it bit for bit against
    the reference's statement sequence on 10^8 random / threshold-hugging / denormal / huge
    pairs for the six normalisations in use, and the shared-reciprocal division against the IEEE
    quotient for every mantissa of the denominator with the reciprocal estimate off by -1, 0
    and +1 ulp (the device's own v_rcp_f32 is covered by tools/ubench/divcheck.hip)."""
    exe = str(tmp_path / "verify_md")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-pthread", "-DGZ_EMU",
                    "-I" + os.path.join(ROOT, "guetzli_amd", "csrc"), "-I" + os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "cpp", "verify_malta_diff.cc"), "-o", exe], check=True)
    out = subprocess.run([exe, "100000000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(" 0 mismatches") == 2, out.stdout


def test_entropy_coder_quantised_values_equal_integer_division(tmp_path):
    """The entropy kernels get `coefficient / q` from one float multiply by 1.0f / q and an integer
    correction (quant_div, gz_kernels_entropy.h).  tests/cpp/verify_quant_div.cc compares it with
    C++'s int division for every dividend an int16 coefficient can be and every 16-bit quantiser."""
    exe = str(tmp_path / "verify_qd")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-pthread", "-DGZ_EMU",
                    "-I" + os.path.join(ROOT, "guetzli_amd", "csrc"), "-I" + os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "cpp", "verify_quant_div.cc"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout, out.stdout


def test_bench_reads_the_chain_valu_floor_from_the_committed_counters():
    """bench.py's roofline.valu: SQ_INSTS_VALU of the chain's kernels per Compare from
    profiles/r03_compare_*_sq_counters.csv over the chip's VALU issue rates (f32 multiply / add:
    2 clocks per wave64 instruction and SIMD; FP64 and the rest: 4)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for size, ms in (("4k", 1.1), ("1080p", 0.36)):
        v = bench.valu_floor(size, ms)
        assert "error" not in v, v
        assert 0.05 < v["floor_ms_f32_rate"] < v["floor_ms_4clk_rate"] < ms
        assert abs(v["floor_ms_4clk_rate"] - 2 * v["floor_ms_f32_rate"]) < 1e-3
        assert v["largest"]["kernel"].startswith("k_malta")
        assert abs(v["frac_of_measured_4clk_rate"] - v["floor_ms_4clk_rate"] / ms) < 1e-3
        # a kernel's launch for the original image at context creation is not part of a Compare:
        # the per-Compare count is below calls x average / Compares for the profiled run
        assert v["wave_instructions_per_compare"] < (470e6 if size == "4k" else 135e6)


def test_host_driver_replays_a_960x540_encode_recorded_on_the_gpu(tmp_path, monkeypatch):
    """The host search driver at a real image size without a GPU: tests/replay's shim stands in for
    the device with the results an MI355X logged for this encode (tools/record_replay.py: original
    coefficients, phase A's candidate lists, the distance and the per-block maxima of every
    evaluation, the symbol statistics and scan sizes; 147 iterations) and the driver -- quant-matrix
    bisection, global order with libstdc++'s tie order through the device-partition path, bulk and
    slow steps, the size model -- must arrive at the JPEG the unmodified reference produces for this
    image (jpeg_sha256_reference: oracle/_ref's Process() on the same pixels, 70 s of CPU,
    generated once by tools/record_replay.py's sidecar step)."""
    import hashlib
    import json
    import lzma
    sys.path.insert(0, os.path.join(ROOT, "tests", "replay"))
    import build_replay
    import images
    from guetzli_amd.encoder import HostLibrary
    gold = os.path.join(ROOT, "tests", "golden", "replay")
    meta = json.load(open(os.path.join(gold, "encode_960x540_q95.json")))
    log = str(tmp_path / "encode.log")
    with lzma.open(os.path.join(gold, "encode_960x540_q95.log.xz")) as f, open(log, "wb") as o:
        o.write(f.read())
    monkeypatch.setenv("GZ_REPLAY_MODE", "replay")
    monkeypatch.setenv("GZ_REPLAY_FILE", log)
    host = HostLibrary(build_replay.build_host())
    jpg, info = host.process(images.tiled(meta["width"], meta["height"]), quality=meta["quality"])
    assert len(jpg) == meta["jpeg_bytes"]
    assert hashlib.sha256(jpg).hexdigest() == meta["jpeg_sha256"] == meta["jpeg_sha256_reference"]
    assert info["counters"]["number of iterations"] == meta["iterations"]


def test_huffman_depths_are_the_reference_tree(tmp_path):
    """The search driver's length-limited Huffman depths (guetzli_amd/host/jpeg_writer.cc: one sort
    or the stream's last order repaired, floor leaves as a fill) against the reference's
    CreateHuffmanTree itself (entropy_encode.cc:73-145, from oracle/_ref/libgz_ref.so): random,
    sparse, tie-heavy and too-deep histograms, and streams of slowly drifting ones as phase B's size
    model produces them -- with and without the stream hint."""
    import ctypes as C
    import numpy as np
    import pytest
    refso = os.path.join(ROOT, "oracle", "_ref", "libgz_ref.so")
    if not os.path.exists(refso):
        pytest.skip("oracle/_ref/libgz_ref.so not built")
    from guetzli_amd import build as gzbuild
    host = C.CDLL(gzbuild.build_host())
    host.gzh_huffman_depths.restype = None
    host.gzh_huffman_depths.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    ref = C.CDLL(refso)
    create = getattr(ref, "_ZN7guetzli17CreateHuffmanTreeEPKjmiPNS_11HuffmanTreeEPh")
    create.restype = None
    create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    tree = np.zeros(8 * (2 * 257 + 1), np.uint8)

    def check(counts, stream, what):
        counts = np.ascontiguousarray(counts, np.uint32)
        want = np.zeros(257, np.uint8)
        got = np.zeros(257, np.uint8)
        create(counts.ctypes.data, 257, 16, tree.ctypes.data, want.ctypes.data)
        host.gzh_huffman_depths(counts.ctypes.data, 16, got.ctypes.data, stream)
        assert np.array_equal(want, got), (what, np.nonzero(want != got)[0][:8])

    rng = np.random.default_rng(20260924)

    def ac_like(scale):
        c = np.zeros(257, np.uint32)
        for s in range(256):
            run, size = s >> 4, s & 15
            if size > 10 or (size == 0 and run not in (0, 15)):
                continue
            c[s] = 2 * int(scale * np.exp(-0.55 * run - 0.9 * abs(size - 2)) * rng.uniform(0.5, 1.5))
        c[256] = 1
        return c

    for i in range(300):       # independent histograms, no hint and a (useless) hint
        kind = i % 6
        if kind == 0:
            c = ac_like(10.0 ** rng.uniform(0, 6.5))
        elif kind == 1:
            c = np.where(rng.random(257) < 0.1, rng.integers(1, 50, 257), 0).astype(np.uint32)
        elif kind == 2:
            c = (2 ** rng.integers(0, 24, 257)).astype(np.uint32) * (rng.random(257) < 0.5)
        elif kind == 3:
            c = np.full(257, rng.integers(1, 5), np.uint32) * (rng.random(257) < 0.7)   # ties
        elif kind == 4:
            c = np.zeros(257, np.uint32)
            c[rng.integers(0, 257, rng.integers(1, 4))] = rng.integers(1, 1000)         # 1-3 symbols
        else:
            fib = [1, 1]
            while len(fib) < 40:
                fib.append(fib[-1] + fib[-2])
            c = np.zeros(257, np.uint32)
            c[:40] = fib                                                                  # deepest tree
        check(c, -1, ("single", kind))
        check(c, i % 8, ("single hinted", kind))
    for stream in range(5):    # drifting histograms on one stream: the cached order is repaired
        c = ac_like(10.0 ** rng.uniform(2, 6)).astype(np.int64)
        for step in range(400):
            for _ in range(rng.integers(1, 12)):
                s = int(rng.integers(0, 256))
                if c[s] or rng.random() < 0.05:
                    c[s] = max(0, c[s] + 2 * int(rng.integers(-3, 4)))   # symbols come and go
            check(c.astype(np.uint32), stream, ("stream", stream, step))
            if step % 50 == 0:   # another stream's histogram in between, under the same id
                check(ac_like(1000.0), stream, ("interleaved", stream, step))
