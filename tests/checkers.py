"""ctypes bindings of the two CHECKERS (test infrastructure, never the product):

  * `oracle`  = oracle/libgz_oracle.so  -- the CPU restatement (gz_oracle.cc)
  * `ref`     = oracle/_ref/libgz_ref.so -- the unmodified reference behind a C shim

Both export the same function set with prefixes orc_ / ref_, so a test can be
parametrised over them.  `ref` is None when the prebuilt library is absent.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_P = C.c_void_p
_SIGS = {
    "fdct_block": (None, [_P]),
    "idct_block": (None, [_P, _P]),
    "quantize_block": (C.c_int, [_P, _P]),
    "ycbcr_to_rgb": (None, [_P, C.c_int]),
    "srgb_to_linear_table": (None, [_P]),
    "encode_rgb": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "reconstruct": (None, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "compute_kernel": (C.c_int, [C.c_float, _P, C.c_int]),
    "blur": (None, [_P, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "opsin": (None, [_P, C.c_int, C.c_int, _P]),
    "separate_frequencies": (None, [_P, C.c_int, C.c_int, _P]),
    "mask": (None, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "malta": (None, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                     C.c_double, _P]),
    "diffmap": (C.c_double, [_P, _P, C.c_int, C.c_int, _P]),
    "comparator_create": (_P, [_P, C.c_int, C.c_int, C.c_float]),
    "comparator_destroy": (None, [_P]),
    "comparator_compare": (C.c_float, [_P, _P, _P]),
    "comparator_block_weights": (None, [_P, C.c_int, C.c_int, C.c_double, _P, _P]),
    "comparator_block_mask": (None, [_P, _P]),
    "comparator_compare_block": (C.c_double, [_P, _P, C.c_int, C.c_int]),
    "block_zeroing_orders": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P,
                                       C.c_int]),
    "to_float_pixels": (None, [_P, C.c_int, C.c_int, _P]),
    "block_zeroing_orders_masked": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P,
                                              _P, C.c_int]),
    "comparator_block_weights_factor": (None, [_P, C.c_int, C.c_int, C.c_double, C.c_int, _P, _P]),
    "comparator_compare420": (C.c_float, [_P, _P, _P]),
    "reconstruct420": (None, [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
    "downsample": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
}
_REF_ONLY = {
    "process": (C.c_long, [_P, C.c_int, C.c_int, C.c_float, _P, C.c_long, _P, C.c_long]),
    "write_jpeg": (C.c_long, [_P, C.c_int, C.c_int, _P, _P, C.c_long]),
    "butteraugli_score_for_quality": (C.c_double, [C.c_double]),
    "score_jpeg": (C.c_double, [C.c_double, C.c_int, C.c_double]),
    "dct_double": (None, [_P]),
    "idct_double": (None, [_P]),
    "downsample_plain": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "process_jpeg": (C.c_long, [_P, C.c_long, C.c_float, C.c_int, _P, C.c_long, _P, C.c_long]),
    "write_jpeg420": (C.c_long, [_P, C.c_int, C.c_int, _P, _P, C.c_long]),
    "process_params": (C.c_long, [_P, C.c_long, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_long, _P, C.c_long]),
}
_ORC_ONLY = {
    "dct_double": (None, [_P, C.c_int]),
    "set_downsampled": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
}


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    return a


class Checker:
    def __init__(self, path, prefix, extra=None):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.name = prefix.rstrip("_")
        sigs = dict(_SIGS)
        if extra:
            sigs.update(extra)
        for name, (res, args) in sigs.items():
            try:
                f = getattr(self.lib, prefix + name)
            except AttributeError:
                continue
            f.restype, f.argtypes = res, args
            setattr(self, "_" + name, f)

    def has(self, name):
        return hasattr(self, "_" + name)

    # ---- block path -------------------------------------------------------------
    def fdct_block(self, block):
        b = np.ascontiguousarray(block, np.int16).copy()
        self._fdct_block(_ptr(b))
        return b

    def idct_block(self, block):
        b = np.ascontiguousarray(block, np.int16)
        out = np.zeros(64, np.uint8)
        self._idct_block(_ptr(b), _ptr(out))
        return out

    def quantize_block(self, block, q):
        b = np.ascontiguousarray(block, np.int16).copy()
        qq = np.ascontiguousarray(q, np.int32)
        ch = self._quantize_block(_ptr(b), _ptr(qq))
        return b, ch

    def ycbcr_to_rgb(self, px):
        p = np.ascontiguousarray(px, np.uint8).copy()
        self._ycbcr_to_rgb(_ptr(p), p.size // 3)
        return p

    def srgb_table(self):
        t = np.zeros(256, np.float64)
        self._srgb_to_linear_table(_ptr(t))
        return t

    def encode_rgb(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, _ = rgb.shape
        nb = ((w + 7) // 8) * ((h + 7) // 8)
        co = np.zeros((3, nb, 64), np.int16)
        assert self._encode_rgb(_ptr(rgb), w, h, _ptr(co)) == 0
        return co

    def reconstruct(self, coeffs, w, h, q=None):
        co = np.ascontiguousarray(coeffs, np.int16)
        qq = None if q is None else np.ascontiguousarray(q, np.int32)
        cout = np.zeros_like(co)
        srgb = np.zeros((h, w, 3), np.uint8)
        lin = np.zeros((3, h, w), np.float32)
        self._reconstruct(_ptr(co), w, h, _ptr(qq), _ptr(cout), _ptr(srgb), _ptr(lin))
        return cout, srgb, lin

    # ---- double-precision DCT (a8) and its users -------------------------------
    def dct_double(self, block, inverse=False):
        b = np.ascontiguousarray(block, np.float64).reshape(64).copy()
        if self.prefix == "orc_":
            self._dct_double(_ptr(b), int(inverse))
        else:
            (self._idct_double if inverse else self._dct_double)(_ptr(b))
        return b

    def to_float_pixels(self, coeffs, w, h):
        co = np.ascontiguousarray(coeffs, np.int16)
        out = np.zeros((h, w), np.float32)
        self._to_float_pixels(_ptr(co), w, h, _ptr(out))
        return out

    def downsample_chroma(self, coeffs, w, h, fx, fy):
        """Chroma coefficients after OutputImage::Downsample with sharpen/blur off:
        ToFloatPixels + SetDownsampledCoefficients of components 1 and 2."""
        co = np.ascontiguousarray(coeffs, np.int16).reshape(3, -1, 64)
        nb = ((w + 8 * fx - 1) // (8 * fx)) * ((h + 8 * fy - 1) // (8 * fy))
        u = np.zeros((nb, 64), np.int16)
        v = np.zeros((nb, 64), np.int16)
        if self.prefix == "orc_":
            for c, dst in ((1, u), (2, v)):
                px = self.to_float_pixels(co[c], w, h)
                assert self._set_downsampled(_ptr(px), w, h, fx, fy, _ptr(dst)) == nb
        else:
            assert self._downsample_plain(_ptr(co), w, h, fx, fy, _ptr(u), _ptr(v)) == nb
        return u, v

    # ---- butteraugli ----------------------------------------------------------
    def compute_kernel(self, sigma):
        t = np.zeros(256, np.float32)
        n = self._compute_kernel(sigma, _ptr(t), 256)
        return t[:n].copy()

    def blur(self, plane, sigma, border_ratio):
        p = np.ascontiguousarray(plane, np.float32)
        h, w = p.shape
        out = np.zeros_like(p)
        self._blur(_ptr(p), w, h, sigma, border_ratio, _ptr(out))
        return out

    def opsin(self, rgb):
        p = np.ascontiguousarray(rgb, np.float32)
        _, h, w = p.shape
        out = np.zeros_like(p)
        self._opsin(_ptr(p), w, h, _ptr(out))
        return out

    def separate_frequencies(self, xyb):
        p = np.ascontiguousarray(xyb, np.float32)
        _, h, w = p.shape
        out = np.zeros((10, h, w), np.float32)
        self._separate_frequencies(_ptr(p), w, h, _ptr(out))
        return out

    def mask(self, xyb0, xyb1):
        a = np.ascontiguousarray(xyb0, np.float32)
        b = np.ascontiguousarray(xyb1, np.float32)
        _, h, w = a.shape
        m = np.zeros((3, h, w), np.float32)
        mdc = np.zeros((3, h, w), np.float32)
        self._mask(_ptr(a), _ptr(b), w, h, _ptr(m), _ptr(mdc))
        return m, mdc

    def malta(self, lum0, lum1, lf, w_0gt1, w_0lt1, norm1, acc=None):
        a = np.ascontiguousarray(lum0, np.float32)
        b = np.ascontiguousarray(lum1, np.float32)
        h, w = a.shape
        out = np.zeros((h, w), np.float32) if acc is None else \
            np.ascontiguousarray(acc, np.float32).copy()
        self._malta(_ptr(a), _ptr(b), w, h, int(lf), w_0gt1, w_0lt1, norm1, _ptr(out))
        return out

    def diffmap(self, rgb0, rgb1):
        a = np.ascontiguousarray(rgb0, np.float32)
        b = np.ascontiguousarray(rgb1, np.float32)
        _, h, w = a.shape
        d = np.zeros((h, w), np.float32)
        score = self._diffmap(_ptr(a), _ptr(b), w, h, _ptr(d))
        return d, score

    # ---- guetzli comparator -----------------------------------------------------
    def comparator(self, rgb, target):
        return CheckerComparator(self, rgb, target)

    # ---- reference only ---------------------------------------------------------
    def process(self, rgb, target, want_trace=False, cap=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, _ = rgb.shape
        cap = cap or (w * h * 3 + (1 << 16))
        out = np.zeros(cap, np.uint8)
        tr = C.create_string_buffer(1 << 22) if want_trace else None
        n = self._process(_ptr(rgb), w, h, target, _ptr(out), cap, tr,
                          len(tr) if tr else 0)
        assert 0 <= n <= cap, n
        return out[:n].tobytes(), (tr.value.decode() if tr else None)

    def process_jpeg(self, data, target, clear_metadata=True, want_trace=False):
        """guetzli::Process(params, stats, jpeg_data, &out); None if it returns false."""
        buf = np.frombuffer(data, np.uint8)
        cap = max(4 * len(data), 1 << 20)
        out = np.zeros(cap, np.uint8)
        tr = C.create_string_buffer(1 << 22) if want_trace else None
        n = self._process_jpeg(_ptr(buf), len(data), target, int(clear_metadata), _ptr(out), cap,
                               tr, len(tr) if tr else 0)
        if n < 0:
            return None, None
        assert n <= cap
        return out[:n].tobytes(), (tr.value.decode() if tr else None)

    # ---- YUV 4:2:0 (frame layout: nb luma blocks, nbc Cb, nbc Cr) ----------------
    @staticmethod
    def blocks420(w, h):
        nb = ((w + 7) // 8) * ((h + 7) // 8)
        nbc = ((w + 15) // 16) * ((h + 15) // 16)
        return nb, nbc

    def downsample(self, coeffs, w, h, silver=False):
        """OutputImage::Downsample as Processor::DownsampleImage configures it."""
        co = np.ascontiguousarray(coeffs, np.int16)
        nb, nbc = self.blocks420(w, h)
        out = np.zeros((3 * nb, 64), np.int16)
        n = self._downsample(_ptr(co), w, h, int(silver), _ptr(out))
        return out[:n].copy()

    def reconstruct420(self, coeffs, w, h, q=None, shuffle=0):
        co = np.ascontiguousarray(coeffs, np.int16)
        qq = None if q is None else np.ascontiguousarray(q, np.int32)
        cout = np.zeros_like(co)
        srgb = np.zeros((h, w, 3), np.uint8)
        lin = np.zeros((3, h, w), np.float32)
        self._reconstruct420(_ptr(co), w, h, _ptr(qq), shuffle, _ptr(cout), _ptr(srgb), _ptr(lin))
        return cout, srgb, lin

    def write_jpeg420(self, coeffs, w, h, q):
        co = np.ascontiguousarray(coeffs, np.int16)
        qq = np.ascontiguousarray(q, np.int32)
        cap = w * h * 3 + (1 << 16)
        out = np.zeros(cap, np.uint8)
        n = self._write_jpeg420(_ptr(co), w, h, _ptr(qq), _ptr(out), cap)
        assert 0 <= n <= cap
        return out[:n].tobytes()

    def process_params(self, data, target, w=0, h=0, clear_metadata=True, try_420=False,
                       force_420=False, silver=False, lookahead=3, new_model=True,
                       want_trace=False):
        """guetzli::Process with all of Params; data = uint8 [h][w][3] array or JPEG bytes.
        Returns (jpeg or None, trace)."""
        if isinstance(data, (bytes, bytearray)):
            buf = np.frombuffer(data, np.uint8)
            jl = len(data)
            cap = max(4 * len(data), 1 << 20)
        else:
            buf = np.ascontiguousarray(data, np.uint8)
            h, w, _ = buf.shape
            jl = -1
            cap = w * h * 3 + (1 << 16)
        out = np.zeros(cap, np.uint8)
        tr = C.create_string_buffer(1 << 22) if want_trace else None
        n = self._process_params(_ptr(buf), jl, w, h, target, int(clear_metadata), int(try_420),
                                 int(force_420), int(silver), lookahead, int(new_model),
                                 _ptr(out), cap, tr, len(tr) if tr else 0)
        if n < 0:
            return None, None
        assert n <= cap
        return out[:n].tobytes(), (tr.value.decode() if tr else None)

    def write_jpeg(self, coeffs, w, h, q):
        co = np.ascontiguousarray(coeffs, np.int16)
        qq = np.ascontiguousarray(q, np.int32)
        cap = w * h * 3 + (1 << 16)
        out = np.zeros(cap, np.uint8)
        n = self._write_jpeg(_ptr(co), w, h, _ptr(qq), _ptr(out), cap)
        assert 0 <= n <= cap
        return out[:n].tobytes()


class CheckerComparator:
    def __init__(self, chk, rgb, target):
        self.chk = chk
        self.rgb = np.ascontiguousarray(rgb, np.uint8)
        self.h, self.w, _ = self.rgb.shape
        self.bw, self.bh = (self.w + 7) // 8, (self.h + 7) // 8
        self.handle = chk._comparator_create(_ptr(self.rgb), self.w, self.h, target)

    def close(self):
        if self.handle:
            self.chk._comparator_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def compare(self, coeffs):
        co = np.ascontiguousarray(coeffs, np.int16)
        d = np.zeros((self.h, self.w), np.float32)
        dist = self.chk._comparator_compare(self.handle, _ptr(co), _ptr(d))
        return dist, d

    def block_weights(self, direction, max_block_dist, target_mul, distmap, weights=None):
        d = np.ascontiguousarray(distmap, np.float32)
        wgt = np.zeros(self.bw * self.bh, np.float32) if weights is None else \
            np.ascontiguousarray(weights, np.float32).copy()
        self.chk._comparator_block_weights(self.handle, direction, max_block_dist,
                                           target_mul, _ptr(d), _ptr(wgt))
        return wgt

    def block_mask(self):
        m = np.zeros((3, self.h, self.w), np.float32)
        self.chk._comparator_block_mask(self.handle, _ptr(m))
        return m

    def compare_block(self, coeffs, bx, by):
        co = np.ascontiguousarray(coeffs, np.int16)
        return self.chk._comparator_compare_block(self.handle, _ptr(co), bx, by)

    def compare420(self, coeffs):
        co = np.ascontiguousarray(coeffs, np.int16)
        d = np.zeros((self.h, self.w), np.float32)
        dist = self.chk._comparator_compare420(self.handle, _ptr(co), _ptr(d))
        return dist, d

    def block_weights_factor(self, direction, max_block_dist, target_mul, factor, distmap,
                             weights=None):
        d = np.ascontiguousarray(distmap, np.float32)
        s = 8 * factor
        n = ((self.w + s - 1) // s) * ((self.h + s - 1) // s)
        wgt = np.zeros(n, np.float32) if weights is None else \
            np.ascontiguousarray(weights, np.float32).copy()
        self.chk._comparator_block_weights_factor(self.handle, direction, max_block_dist,
                                                  target_mul, factor, _ptr(d), _ptr(wgt))
        return wgt

    def block_zeroing_orders_masked(self, coeffs, orig, frame420, comp_mask, lookahead=3,
                                    new_model=True):
        co = np.ascontiguousarray(coeffs, np.int16)
        og = np.ascontiguousarray(orig, np.int16)
        f = 2 if (frame420 and comp_mask & 6) else 1
        gn = ((self.w + 8 * f - 1) // (8 * f)) * ((self.h + 8 * f - 1) // (8 * f))
        cap = gn * 192
        off = np.zeros(gn + 1, np.int32)
        idx = np.zeros(cap, np.uint8)
        err = np.zeros(cap, np.float32)
        n = self.chk._block_zeroing_orders_masked(self.handle, _ptr(co), _ptr(og), int(frame420),
                                                  comp_mask, lookahead, int(new_model), _ptr(off),
                                                  _ptr(idx), _ptr(err), cap)
        assert n >= 0
        return off, idx[:n].copy(), err[:n].copy()

    def block_zeroing_orders(self, coeffs, orig, lookahead=3, new_model=True):
        co = np.ascontiguousarray(coeffs, np.int16)
        og = np.ascontiguousarray(orig, np.int16)
        nb = self.bw * self.bh
        cap = nb * 192
        off = np.zeros(nb + 1, np.int32)
        idx = np.zeros(cap, np.uint8)
        err = np.zeros(cap, np.float32)
        n = self.chk._block_zeroing_orders(self.handle, _ptr(co), _ptr(og), lookahead,
                                           int(new_model), _ptr(off), _ptr(idx),
                                           _ptr(err), cap)
        assert n >= 0
        return off, idx[:n].copy(), err[:n].copy()


def build_oracle():
    """(Re)build oracle/libgz_oracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def _load():
    so = os.path.join(ORACLE_DIR, "libgz_oracle.so")
    src = os.path.join(ORACLE_DIR, "gz_oracle.cc")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    orc = Checker(so, "orc_", _ORC_ONLY)
    refso = os.path.join(ORACLE_DIR, "_ref", "libgz_ref.so")
    rf = Checker(refso, "ref_", _REF_ONLY) if os.path.exists(refso) else None
    return orc, rf


oracle, ref = _load()


def bits(a):
    """View a float array as raw bits so that comparisons are bit-exact (and NaN-safe)."""
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind == "f":
        bad = bits(a) != bits(b)
    else:
        bad = a != b
    if bad.any():
        idx = np.argwhere(bad)
        i = tuple(idx[0])
        raise AssertionError(
            f"{what}: {bad.sum()} of {bad.size} elements differ; first at {i}: "
            f"{a[i]!r} vs {b[i]!r}")
