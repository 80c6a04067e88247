"""Out-of-memory paths of the C ABI and of the host driver above it (VERDICT r5 "HIP-failure paths are untested").

The CPU emulation's hipMalloc / hipHostMalloc (tests/emu/hip_emu.h) count every allocation, can be told to fail the
n-th one from now on with hipErrorOutOfMemory, and list what is still allocated.  Checked here:
  * gz_create with EVERY one of its allocations failing in turn: NULL + GZ_E_NOMEM, nothing left allocated (device
    blocks, page-locked blocks, events), and the next gz_create succeeds;
  * a whole encode through the host driver with allocations failing at sampled points of its life (context creation,
    block search, phase B's order, the entropy coder's buffers, staging): the encode fails with an error -- no crash,
    nothing left allocated -- and the encode after it produces the reference's bytes.
(The emulation build allocates directly instead of through the product's pools, so "still allocated" is exact.)
CPU only."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import pytest

import images
from checkers import ref

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import build_emu  # noqa: E402

GZ_E_NOMEM = -5


@pytest.fixture(scope="module")
def emu():
    lib = C.CDLL(build_emu.build())
    lib.gz_create.restype = C.c_void_p
    lib.gz_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.POINTER(C.c_int)]
    lib.gz_destroy.argtypes = [C.c_void_p]
    lib.gz_emu_fail_alloc.argtypes = [C.c_long]
    lib.gz_emu_alloc_calls.restype = C.c_long
    lib.gz_emu_live.argtypes = [C.POINTER(C.c_long)] * 4
    return lib


def live(lib):
    v = [C.c_long() for _ in range(4)]
    lib.gz_emu_live(*[C.byref(x) for x in v])
    return tuple(x.value for x in v)   # device bytes, page-locked bytes, blocks, events


def test_gz_create_survives_every_allocation_failing(emu):
    rgb = np.ascontiguousarray(images.crop(40, 32, 100, 60))
    err = C.c_int(0)
    base = live(emu)
    before = emu.gz_emu_alloc_calls()
    ctx = emu.gz_create(0, 40, 32, rgb.ctypes.data, 0.971769, C.byref(err))
    assert ctx and err.value == 0
    n_allocs = emu.gz_emu_alloc_calls() - before
    assert n_allocs >= 15, n_allocs          # the arena, the coefficient arrays, tables, blur scales, staging ...
    emu.gz_destroy(ctx)
    assert live(emu) == base, "a clean create / destroy leaves allocations behind"
    for n in range(n_allocs):
        emu.gz_emu_fail_alloc(n)
        ctx = emu.gz_create(0, 40, 32, rgb.ctypes.data, 0.971769, C.byref(err))
        assert not ctx, f"gz_create succeeded although allocation {n} failed"
        assert err.value == GZ_E_NOMEM, (n, err.value)
        assert live(emu) == base, f"allocation {n} failing leaves {live(emu)} (device bytes, host bytes, blocks, events)"
        emu.gz_emu_fail_alloc(-1)
        ctx = emu.gz_create(0, 40, 32, rgb.ctypes.data, 0.971769, C.byref(err))
        assert ctx and err.value == 0, f"gz_create after the failure of allocation {n}"
        emu.gz_destroy(ctx)
        assert live(emu) == base


@pytest.mark.skipif(ref is None, reason="oracle/_ref/libgz_ref.so not built")
def test_an_encode_survives_allocations_failing_anywhere(emu):
    """The host driver over the same library: allocation k of an encode fails -> Process fails cleanly."""
    from guetzli_amd.encoder import HostLibrary
    host = HostLibrary(build_emu.build_host())
    rgb = np.ascontiguousarray(images.crop(40, 32, 100, 60))
    exp = hashlib.sha256(ref.process(rgb, ref._butteraugli_score_for_quality(90.0))[0]).hexdigest()
    base = live(emu)
    before = emu.gz_emu_alloc_calls()
    jpg, _ = host.process(rgb, quality=90.0)
    assert hashlib.sha256(jpg).hexdigest() == exp
    total = emu.gz_emu_alloc_calls() - before
    assert live(emu) == base, "a clean encode leaves allocations behind"
    assert total >= 40, total
    # every allocation of the first 60 (context, first calls of every entry point), then every seventh
    points = sorted(set(range(min(60, total))) | set(range(60, total, 7)) | {total - 1})
    failed = 0
    for k in points:
        emu.gz_emu_fail_alloc(k)
        try:
            out, _ = host.process(rgb, quality=90.0)
            # an allocation that the encode can do without (a cache, an optional buffer) may be survived -- then
            # the bytes must still be the reference's
            assert hashlib.sha256(out).hexdigest() == exp, f"allocation {k} failed and the output changed"
        except RuntimeError as e:
            failed += 1
            assert "memory" in str(e).lower() or "hip" in str(e).lower() or "failed" in str(e).lower(), str(e)
        finally:
            emu.gz_emu_fail_alloc(-1)
        assert live(emu) == base, f"allocation {k} of {total} failing leaves {live(emu)} behind"
    assert failed >= len(points) // 2, (failed, len(points))
    jpg, _ = host.process(rgb, quality=90.0)
    assert hashlib.sha256(jpg).hexdigest() == exp
