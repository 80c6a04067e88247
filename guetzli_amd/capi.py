"""ctypes binding of the C ABI in include/guetzli_amd.h.

`load()` opens guetzli_amd/libguetzli_amd.so -- the hipcc-built gfx950 library -- and
nothing else: there is no CPU implementation behind this module, and a missing library or
a missing GPU is an error, not a fallback.  (The test-suite's CPU emulation build of the
kernel sources is loaded by tests through `Library(path)` explicitly; the package never
does that.)
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libguetzli_amd.so")

GZ_OK = 0
ERRORS = {-1: "GZ_E_ARG", -2: "GZ_E_NO_DEVICE", -3: "GZ_E_HIP", -4: "GZ_E_STATE",
          -5: "GZ_E_NOMEM"}

_P = C.c_void_p
_I = C.c_int


class GzConfig(C.Structure):
    """gz_config of include/guetzli_amd.h."""
    _fields_ = [("struct_size", _I), ("blur_packed", _I), ("tile_rows", _I), ("single_stream", _I),
                ("store_distmap", _I), ("side_small", _I), ("malta_pad_bytes", _I), ("patch_reconstruct", _I), ("opsin_ahead", _I)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

# name -> (restype, argtypes); mirrors include/guetzli_amd.h one to one
SIGNATURES = {
    "gz_abi_version": (_I, []),
    "gz_device_count": (_I, []),
    "gz_strerror": (C.c_char_p, [_I]),
    "gz_last_error": (C.c_char_p, [_P]),
    "gz_hint_images_in_flight": (None, [_I]),
    "gz_device_pci_bus_id": (_I, [_I, _P, _I]),
    "gz_config_from_environment": (_I, [_P]),
    "gz_get_config": (_I, [_P, _P]),
    "gz_set_config": (_I, [_P, _P]),
    "gz_create": (_P, [_I, _I, _I, _P, C.c_float, C.POINTER(_I)]),
    "gz_destroy": (None, [_P]),
    "gz_set_rgb": (_I, [_P, _P]),
    "gz_synchronize": (_I, [_P]),
    "gz_set_stream": (_I, [_P, _P]),
    "gz_encode_rgb": (_I, [_P, _P]),
    "gz_encode_rgb_only": (_I, [_I, _P, _I, _I, _P]),
    "gz_set_orig_coeffs": (_I, [_P, _P]),
    "gz_set_orig_coeffs_420": (_I, [_P, _P]),
    "gz_downsample": (_I, [_P, _P]),
    "gz_downsample_planes": (_I, [_P, _P, _P, _P, _P]),
    "gz_frame_layout": (_I, [_P, _P, _P, _P]),
    "gz_quantize": (_I, [_P, _P, _P]),
    "gz_set_coeffs": (_I, [_P, _P]),
    "gz_set_coeff_blocks": (_I, [_P, _P, _I, _P]),
    "gz_get_coeffs": (_I, [_P, _P]),
    "gz_reconstruct": (_I, [_P, _P, _P]),
    "gz_trim_pool": (_I, []),
    "gz_compare": (_I, [_P, _P, _P, _P]),
    "gz_compare_begin": (_I, [_P]),
    "gz_compare_end": (_I, [_P, _P]),
    "gz_compare_enqueue": (_I, [_P, _I]),
    "gz_last_distance": (_I, [_P, _P]),
    "gz_time_compare": (_I, [_P, _I, _P]),
    "gz_block_weights": (_I, [_P, _I, _I, C.c_double, _I, _P]),
    "gz_block_weights_factor": (_I, [_P, _I, _I, C.c_double, _I, _I, _P]),
    "gz_block_zeroing_orders": (_I, [_P, _I, _I, _P, _P, _P, _I]),
    "gz_block_zeroing_orders_masked": (_I, [_P, _I, _I, _I, _P, _P, _P, _I]),
    "gz_compare_blocks": (_I, [_P, _I, _P, _P, _P]),
    "gz_compare_block_pixels": (_I, [_P, _I, _P, _P, _P]),
    "gz_set_frame": (_I, [_P, _I]),
    "gz_search_evaluations": (_I, [_P, _P]),
    "gz_compare_counters": (_I, [_P]),
    "gz_rank_zeroing_candidates": (_I, [_P, _P, _I, _I, _P, _P]),
    "gz_probe_rank_sort": (_I, [_I, _P, _P, _I, _P]),
    "gz_order_build": (_I, [_P, _I, _P, _P, _P, _I, C.c_float, _P, _P, _P]),
    "gz_order_reset": (_I, [_P]),
    "gz_order_build_auto": (_I, [_P, _I, _I, C.c_double, _I, _P, _I, C.c_float, _P, _P, _P]),
    "gz_order_build_auto_begin": (_I, [_P, _I, _I, C.c_double, _I, _P, _I, C.c_float]),
    "gz_order_build_auto_descend_begin": (_I, [_P, _I, _I, C.c_double, _I, _P, _I, C.c_float, C.c_float, C.c_uint64, _I]),
    "gz_order_build_auto_end": (_I, [_P, _P, _P, _P]),
    "gz_order_advance": (_I, [_P, C.c_float, _I]),
    "gz_apply_coeff_edits": (_I, [_P, _P, _P, _I]),
    "gz_apply_candidate_steps": (_I, [_P, _I, _P, _P, _I]),
    "gz_steps_histogram_delta": (_I, [_P, _P]),
    "gz_order_upload": (_I, [_P, _P, C.c_uint64]),
    "gz_order_partition": (_I, [_P, C.c_uint64, C.c_uint64, _P]),
    "gz_order_fetch": (_I, [_P, C.c_uint64, C.c_uint64, _P]),
    "gz_order_host_mirror": (_I, [_P, C.c_uint64, _P]),
    "gz_order_exported": (_I, [_P, _P]),
    "gz_order_descend": (_I, [_P, C.c_uint64, C.c_uint64, _I, _P, _P]),
    "gz_order_descend_begin": (_I, [_P, C.c_float, C.c_uint64, _I]),
    "gz_order_descend_end": (_I, [_P, _P, _I, _P, _P]),
    "gz_jpeg_histograms": (_I, [_P, _P, _P]),
    "gz_jpeg_histograms_ncomp": (_I, [_P, _P, _I, _P]),
    "gz_jpeg_scan": (_I, [_P, _I, _P, _P, _P]),
    "gz_jpeg_scan_begin": (_I, [_P, _I, _P, _P]),
    "gz_jpeg_scan_end": (_I, [_P, _P]),
    "gz_jpeg_scan_keep": (_I, [_P]),
    "gz_jpeg_scan_bits": (_I, [_P, _P, _P]),
    "gz_jpeg_scan_bytes": (_I, [_P, _I, _P, C.c_size_t, _P]),
    "gz_probe_blur": (_I, [_P, _P, C.c_float, C.c_float, _P]),
    "gz_probe_opsin": (_I, [_P, _P, _P]),
    "gz_probe_separate_frequencies": (_I, [_P, _P, _P]),
    "gz_probe_diffmap": (_I, [_P, _P, _P, _P, _P]),
    "gz_probe_mask": (_I, [_P, _P, _P, _P, _P]),
    "gz_probe_idct_blocks": (_I, [_I, _P, _I, _P]),
    "gz_probe_fdct_blocks": (_I, [_I, _P, _I]),
    "gz_probe_arith": (_I, [_I, _I, _P, _P, _P, _P, _I]),
    "gz_dct_double_blocks": (_I, [_I, _P, _I, _I]),
    "gz_component_to_float_pixels": (_I, [_I, _P, _I, _I, _P]),
    "gz_component_set_downsampled": (_I, [_I, _P, _I, _I, _I, _I, _P]),
}


class GuetzliAmdError(RuntimeError):
    pass


def _ptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "need a C-contiguous array"
    return a.ctypes.data


class Library:
    def __init__(self, path=DEFAULT_LIB):
        if not os.path.exists(path):
            raise GuetzliAmdError(
                f"{path} not found: build it with `python -m guetzli_amd.build` (hipcc, "
                "gfx950).  There is no CPU fallback.")
        self.path = path
        self.lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(self.lib, name)   # AttributeError if the ABI is incomplete
            f.restype, f.argtypes = res, args

    def check(self, rc, ctx=None):
        if rc != GZ_OK:
            msg = self.lib.gz_strerror(rc).decode()
            if ctx:
                msg += ": " + self.lib.gz_last_error(ctx).decode()
            raise GuetzliAmdError(f"{ERRORS.get(rc, rc)} ({msg})")

    def device_count(self):
        return self.lib.gz_device_count()

    def device_pci_bus_id(self, device=0):
        buf = C.create_string_buffer(32)
        self.check(self.lib.gz_device_pci_bus_id(device, buf, 32))
        return buf.value.decode()

    def config_from_environment(self):
        cfg = GzConfig()
        self.check(self.lib.gz_config_from_environment(C.byref(cfg)))
        return cfg

    def compare_counters(self, all=False):
        """(Compares that skipped the full reconstruction, of them checked against one, Compares in all) since load;
        all=True: + (Compares that found their opsin image in place, of them checked)."""
        n = np.zeros(5, np.uint64)
        self.check(self.lib.gz_compare_counters(_ptr(n)))
        return tuple(int(x) for x in (n if all else n[:3]))

    # ---- context-free probes ----
    def idct_blocks(self, blocks, device=0):
        b = np.ascontiguousarray(blocks, np.int16).reshape(-1, 64)
        out = np.zeros(b.shape, np.uint8)
        self.check(self.lib.gz_probe_idct_blocks(device, _ptr(b), b.shape[0], _ptr(out)))
        return out

    def fdct_blocks(self, blocks, device=0):
        b = np.ascontiguousarray(blocks, np.int16).reshape(-1, 64).copy()
        self.check(self.lib.gz_probe_fdct_blocks(device, _ptr(b), b.shape[0]))
        return b

    def encode_rgb_only(self, rgb, device=0):
        """EncodeRGBToJpeg's transform (q = 1) without a context, any size >= 1x1."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, _ = rgb.shape
        nb = ((w + 7) // 8) * ((h + 7) // 8)
        out = np.zeros((3, nb, 64), np.int16)
        self.check(self.lib.gz_encode_rgb_only(device, _ptr(rgb), w, h, _ptr(out)))
        return out

    def dct_double_blocks(self, blocks, inverse=False, device=0):
        """ComputeBlockDCTDouble / ComputeBlockIDCTDouble (dct_double.cc:76-85) per block."""
        b = np.ascontiguousarray(blocks, np.float64).reshape(-1, 64).copy()
        self.check(self.lib.gz_dct_double_blocks(device, _ptr(b), b.shape[0], int(inverse)))
        return b

    def component_to_float_pixels(self, coeffs, w, h, device=0):
        """OutputImageComponent::ToFloatPixels (output_image.cc:99-121), stride 1."""
        nb = ((w + 7) // 8) * ((h + 7) // 8)
        co = np.ascontiguousarray(coeffs, np.int16)
        assert co.size == nb * 64
        out = np.zeros((h, w), np.float32)
        self.check(self.lib.gz_component_to_float_pixels(device, _ptr(co), w, h, _ptr(out)))
        return out

    def component_set_downsampled(self, pixels, fx, fy, device=0):
        """SetDownsampledCoefficients (output_image.cc:265-300) of an h x w float plane."""
        px = np.ascontiguousarray(pixels, np.float32)
        h, w = px.shape
        nb = ((w + 8 * fx - 1) // (8 * fx)) * ((h + 8 * fy - 1) // (8 * fy))
        out = np.zeros((nb, 64), np.int16)
        self.check(self.lib.gz_component_set_downsampled(device, _ptr(px), w, h, fx, fy,
                                                         _ptr(out)))
        return out

    def arith(self, op, a, b=None, c=None, device=0):
        dt = np.float64 if op in (2, 3, 5, 6) else np.float32
        odt = np.float64 if op in (2, 3, 5) else np.float32
        a = np.ascontiguousarray(a, dt)
        b = None if b is None else np.ascontiguousarray(b, dt)
        c = None if c is None else np.ascontiguousarray(c, dt)
        out = np.zeros(a.shape, odt)
        self.check(self.lib.gz_probe_arith(device, op, _ptr(a), _ptr(b), _ptr(c), _ptr(out),
                                           a.size))
        return out

    def context(self, rgb, target, device=0):
        return Context(self, rgb, target, device)


class Context:
    """One (image, GPU) context == one guetzli::ButteraugliComparator + OutputImage."""

    def __init__(self, library, rgb, target, device=0):
        self.L = library
        self.rgb = np.ascontiguousarray(rgb, np.uint8)
        self.h, self.w, ch = self.rgb.shape
        assert ch == 3
        self.bw, self.bh = (self.w + 7) // 8, (self.h + 7) // 8
        self.nb = self.bw * self.bh
        # the frame: 4:4:4 (coefficient arrays [3][nb][64]) until downsample() /
        # set_orig_coeffs_420() make it 4:2:0 (flat [nb + 2*nbc][64]: Y, Cb, Cr)
        self.cfac = 1
        self.cbw, self.cbh = (self.w + 15) // 16, (self.h + 15) // 16
        self.nbc = self.cbw * self.cbh
        err = C.c_int(0)
        self.handle = library.lib.gz_create(device, self.w, self.h, _ptr(self.rgb),
                                            float(target), C.byref(err))
        if not self.handle:
            library.check(err.value or -3)

    def close(self):
        if getattr(self, "handle", None):
            self.L.lib.gz_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        self.L.check(rc, self.handle)

    def _coeff_buf(self, cfac=None):
        if (cfac or self.cfac) == 2:
            return np.zeros((self.nb + 2 * self.nbc, 64), np.int16)
        return np.zeros((3, self.nb, 64), np.int16)

    @property
    def nblk(self):
        return self.nb + 2 * self.nbc if self.cfac == 2 else 3 * self.nb

    def split420(self, coeffs):
        """(Y [nb][64], Cb [nbc][64], Cr [nbc][64]) views of a 4:2:0 coefficient array."""
        co = np.asarray(coeffs).reshape(-1, 64)
        return co[:self.nb], co[self.nb:self.nb + self.nbc], co[self.nb + self.nbc:]

    def frame_layout(self):
        f, a, b = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.L.lib.gz_frame_layout(self.handle, C.byref(f), C.byref(a), C.byref(b)))
        return f.value, a.value, b.value

    def set_rgb(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert rgb.shape == (self.h, self.w, 3)
        self.rgb = rgb
        self._chk(self.L.lib.gz_set_rgb(self.handle, _ptr(rgb)))

    def synchronize(self):
        self._chk(self.L.lib.gz_synchronize(self.handle))

    def get_config(self):
        cfg = GzConfig()
        self._chk(self.L.lib.gz_get_config(self.handle, C.byref(cfg)))
        return cfg

    def set_config(self, **fields):
        """Replaces fields of the context's gz_config (blur_packed, tile_rows, single_stream, ...)."""
        cfg = self.get_config()
        for k, v in fields.items():
            assert k in dict(GzConfig._fields_), k
            setattr(cfg, k, v)
        self._chk(self.L.lib.gz_set_config(self.handle, C.byref(cfg)))
        return cfg

    def set_stream(self, stream_ptr):
        self._chk(self.L.lib.gz_set_stream(self.handle, stream_ptr))

    def encode_rgb(self, download=True):
        self.cfac = 1
        out = self._coeff_buf() if download else None
        self._chk(self.L.lib.gz_encode_rgb(self.handle, _ptr(out)))
        return out

    def set_orig_coeffs(self, coeffs):
        co = np.ascontiguousarray(coeffs, np.int16)
        assert co.size == 3 * self.nb * 64
        self._chk(self.L.lib.gz_set_orig_coeffs(self.handle, _ptr(co)))
        self.cfac = 1   # (only once the library has switched its frame)

    def set_orig_coeffs_420(self, coeffs):
        co = np.ascontiguousarray(coeffs, np.int16)
        assert co.size == (self.nb + 2 * self.nbc) * 64
        self._chk(self.L.lib.gz_set_orig_coeffs_420(self.handle, _ptr(co)))
        self.cfac = 2

    def downsample(self, download=True):
        """OutputImage::Downsample on the original: the frame becomes 4:2:0."""
        out = self._coeff_buf(2) if download else None
        self._chk(self.L.lib.gz_downsample(self.handle, _ptr(out)))
        self.cfac = 2   # (a failed call leaves the 4:4:4 layout in place on both sides)
        return out

    def downsample_planes(self, y, u, v, download=True):
        """The silver-screen branch of OutputImage::Downsample, given RGBToYUV420's planes."""
        pl = [np.ascontiguousarray(p, np.float32) for p in (y, u, v)]
        assert all(p.size == self.w * self.h for p in pl)
        out = self._coeff_buf(2) if download else None
        self._chk(self.L.lib.gz_downsample_planes(self.handle, _ptr(pl[0]), _ptr(pl[1]), _ptr(pl[2]), _ptr(out)))
        self.cfac = 2
        return out

    def quantize(self, q=None, download=True):
        qq = None if q is None else np.ascontiguousarray(q, np.int32)
        out = self._coeff_buf() if download else None
        self._chk(self.L.lib.gz_quantize(self.handle, _ptr(qq), _ptr(out)))
        return out

    def set_coeffs(self, coeffs):
        co = np.ascontiguousarray(coeffs, np.int16)
        assert co.size == self.nblk * 64
        self._chk(self.L.lib.gz_set_coeffs(self.handle, _ptr(co)))

    def set_coeff_blocks(self, block_index, blocks):
        bi = np.ascontiguousarray(block_index, np.int32)
        bl = np.ascontiguousarray(blocks, np.int16)
        assert bl.size == bi.size * 192
        self._chk(self.L.lib.gz_set_coeff_blocks(self.handle, _ptr(bi), bi.size, _ptr(bl)))

    def get_coeffs(self):
        out = self._coeff_buf()
        self._chk(self.L.lib.gz_get_coeffs(self.handle, _ptr(out)))
        return out

    def reconstruct(self):
        srgb = np.zeros((self.h, self.w, 3), np.uint8)
        lin = np.zeros((3, self.h, self.w), np.float32)
        self._chk(self.L.lib.gz_reconstruct(self.handle, _ptr(srgb), _ptr(lin)))
        return srgb, lin

    def compare(self, want_distmap=True, want_block_max=True):
        dist = np.zeros(1, np.float32)
        dm = np.zeros((self.h, self.w), np.float32) if want_distmap else None
        bm = np.zeros(self.nb, np.float32) if want_block_max else None
        self._chk(self.L.lib.gz_compare(self.handle, _ptr(dist), _ptr(dm), _ptr(bm)))
        return float(dist[0]), dm, bm

    def compare_begin(self):
        self._chk(self.L.lib.gz_compare_begin(self.handle))

    def compare_end(self):
        d = np.zeros(1, np.float32)
        self._chk(self.L.lib.gz_compare_end(self.handle, _ptr(d)))
        return float(d[0])

    def compare_enqueue(self, iters):
        self._chk(self.L.lib.gz_compare_enqueue(self.handle, iters))

    def last_distance(self):
        d = np.zeros(1, np.float32)
        self._chk(self.L.lib.gz_last_distance(self.handle, _ptr(d)))
        return float(d[0])

    def time_compare(self, iters):
        ms = np.zeros(1, np.float32)
        self._chk(self.L.lib.gz_time_compare(self.handle, iters, _ptr(ms)))
        return float(ms[0])

    def block_weights(self, direction, max_block_dist, target_mul, use_distmap=True,
                      weights=None):
        wgt = np.zeros(self.nb, np.float32) if weights is None else \
            np.ascontiguousarray(weights, np.float32).copy()
        self._chk(self.L.lib.gz_block_weights(self.handle, direction, max_block_dist,
                                              target_mul, int(use_distmap), _ptr(wgt)))
        return wgt

    def block_weights_factor(self, direction, max_block_dist, target_mul, factor,
                             use_distmap=True, weights=None):
        n = self.nbc if factor == 2 else self.nb
        wgt = np.zeros(n, np.float32) if weights is None else \
            np.ascontiguousarray(weights, np.float32).copy()
        self._chk(self.L.lib.gz_block_weights_factor(self.handle, direction, max_block_dist,
                                                     target_mul, int(use_distmap), factor, _ptr(wgt)))
        return wgt

    def block_zeroing_orders(self, lookahead=3, new_model=True, comp_mask=7):
        gn = self.nbc if (self.cfac == 2 and comp_mask == 6) else self.nb
        self.search_blocks = gn
        cap = gn * 189
        off = np.zeros(gn + 1, np.int32)
        idx = np.zeros(cap, np.uint8)
        err = np.zeros(cap, np.float32)
        self._chk(self.L.lib.gz_block_zeroing_orders_masked(self.handle, comp_mask, lookahead,
                                                            int(new_model), _ptr(off), _ptr(idx),
                                                            _ptr(err), cap))
        n = int(off[-1])
        return off, idx[:n].copy(), err[:n].copy()

    def search_evaluations(self):
        n = np.zeros(1, np.uint64)
        self._chk(self.L.lib.gz_search_evaluations(self.handle, _ptr(n)))
        return int(n[0])

    def compare_blocks(self, block_xy, coeffs):
        """SwitchBlock + CompareBlock for n (block position, 3x64 coefficients) pairs."""
        xy = np.ascontiguousarray(block_xy, np.int32).reshape(-1, 2)
        co = np.ascontiguousarray(coeffs, np.int16).reshape(-1, 3, 64)
        assert len(xy) == len(co)
        out = np.zeros(len(xy), np.float64)
        self._chk(self.L.lib.gz_compare_blocks(self.handle, len(xy), _ptr(xy), _ptr(co), _ptr(out)))
        return out

    def compare_block_pixels(self, block_xy, ycc):
        """SwitchBlock + CompareBlock for n 8x8 windows given by their YCbCr pixels (n x 3 x 64 u8)."""
        xy = np.ascontiguousarray(block_xy, np.int32).reshape(-1, 2)
        px = np.ascontiguousarray(ycc, np.uint8).reshape(-1, 3, 64)
        assert len(xy) == len(px)
        out = np.zeros(len(xy), np.float64)
        self._chk(self.L.lib.gz_compare_block_pixels(self.handle, len(xy), _ptr(xy), _ptr(px), _ptr(out)))
        return out

    def set_frame(self, chroma_factor):
        self._chk(self.L.lib.gz_set_frame(self.handle, int(chroma_factor)))
        self.cfac = int(chroma_factor)

    # ---- global candidate order of phase B ----
    ORDER_DTYPE = np.dtype([("block", np.int32), ("val", np.float32)])

    def order_build(self, direction, next_cand, max_block_error, block_weight, limit=None):
        nc = np.ascontiguousarray(next_cand, np.int32)
        me = np.ascontiguousarray(max_block_error, np.float32)
        bw = np.ascontiguousarray(block_weight, np.float32)
        assert nc.size == me.size == bw.size == getattr(self, "search_blocks", self.nb)
        total, below = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        btc = np.zeros(1, np.int32)
        self._chk(self.L.lib.gz_order_build(self.handle, direction, _ptr(nc), _ptr(me), _ptr(bw),
                                            int(limit is not None), float(limit or 0.0),
                                            _ptr(total), _ptr(btc), _ptr(below)))
        return int(total[0]), int(btc[0]), int(below[0])

    def order_reset(self):
        self._chk(self.L.lib.gz_order_reset(self.handle))

    def order_build_auto(self, direction, max_block_dist, target_mul, use_distmap, next_cand,
                         limit=None):
        nc = np.ascontiguousarray(next_cand, np.int32)
        assert nc.size == getattr(self, "search_blocks", self.nb)
        total, below = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        btc = np.zeros(1, np.int32)
        self._chk(self.L.lib.gz_order_build_auto(self.handle, direction, max_block_dist,
                                                 target_mul, int(use_distmap), _ptr(nc),
                                                 int(limit is not None), float(limit or 0.0),
                                                 _ptr(total), _ptr(btc), _ptr(below)))
        return int(total[0]), int(btc[0]), int(below[0])

    def order_build_auto_begin(self, direction, max_block_dist, target_mul, use_distmap, next_cand,
                               limit=None):
        nc = np.ascontiguousarray(next_cand, np.int32)
        assert nc.size == getattr(self, "search_blocks", self.nb)
        self._chk(self.L.lib.gz_order_build_auto_begin(self.handle, direction, max_block_dist, target_mul,
                                                       int(use_distmap), _ptr(nc), int(limit is not None),
                                                       float(limit or 0.0)))

    def order_build_auto_end(self):
        """(total, blocks_to_change, below)."""
        total, below = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        btc = np.zeros(1, np.int32)
        self._chk(self.L.lib.gz_order_build_auto_end(self.handle, _ptr(total), _ptr(btc), _ptr(below)))
        return int(total[0]), int(btc[0]), int(below[0])

    def order_advance(self, val_threshold, direction):
        self._chk(self.L.lib.gz_order_advance(self.handle, float(val_threshold), direction))

    def apply_coeff_edits(self, pos, val):
        p = np.ascontiguousarray(pos, np.int32)
        v = np.ascontiguousarray(val, np.int16)
        assert p.size == v.size
        self._chk(self.L.lib.gz_apply_coeff_edits(self.handle, _ptr(p), _ptr(v), p.size))

    def apply_candidate_steps(self, direction, blocks, counts):
        b = np.ascontiguousarray(blocks, np.int32)
        n = np.ascontiguousarray(counts, np.int32)
        assert b.size == n.size
        self._chk(self.L.lib.gz_apply_candidate_steps(self.handle, direction, _ptr(b), _ptr(n),
                                                      b.size))

    def steps_histogram_delta(self):
        """AC statistics change of the last apply_candidate_steps, int64 [3][256]."""
        d = np.zeros((3, 256), np.int32)
        self._chk(self.L.lib.gz_steps_histogram_delta(self.handle, _ptr(d)))
        return d.astype(np.int64)

    def order_upload(self, entries):
        e = np.ascontiguousarray(entries, self.ORDER_DTYPE)
        self._chk(self.L.lib.gz_order_upload(self.handle, _ptr(e), e.size))

    def order_partition(self, lo, hi):
        cut = np.zeros(1, np.uint64)
        self._chk(self.L.lib.gz_order_partition(self.handle, lo, hi, _ptr(cut)))
        return int(cut[0])

    def order_fetch(self, lo, hi):
        out = np.zeros(hi - lo, self.ORDER_DTYPE)
        self._chk(self.L.lib.gz_order_fetch(self.handle, lo, hi, _ptr(out)))
        return out

    # ---- entropy coding of the candidate ----
    def jpeg_histograms(self, q, ncomp=3):
        """(DC, AC) x component x symbol occurrence counts, uint32 [2][3][256]."""
        qq = np.ascontiguousarray(q, np.int32)
        assert qq.shape == (3, 64)
        counts = np.zeros((2, 3, 256), np.uint32)
        self._chk(self.L.lib.gz_jpeg_histograms_ncomp(self.handle, _ptr(qq), ncomp, _ptr(counts)))
        return counts

    def jpeg_scan(self, ncomp, depth, code):
        """Encodes the scan on the device; returns its exact (stuffed) size in bytes."""
        d = np.ascontiguousarray(depth, np.uint8)
        cd = np.ascontiguousarray(code, np.uint16)
        assert d.shape == (2, 3, 256) and cd.shape == (2, 3, 256)
        n = np.zeros(1, np.uint64)
        self._chk(self.L.lib.gz_jpeg_scan(self.handle, ncomp, _ptr(d), _ptr(cd), _ptr(n)))
        return int(n[0])

    def jpeg_scan_begin(self, ncomp, depth, code):
        """jpeg_scan in two halves: enqueue (entropy stream) ..."""
        d = np.ascontiguousarray(depth, np.uint8)
        cd = np.ascontiguousarray(code, np.uint16)
        assert d.shape == (2, 3, 256) and cd.shape == (2, 3, 256)
        self._chk(self.L.lib.gz_jpeg_scan_begin(self.handle, ncomp, _ptr(d), _ptr(cd)))

    def jpeg_scan_end(self):
        """... and collect: the scan's exact (stuffed) size in bytes."""
        n = np.zeros(1, np.uint64)
        self._chk(self.L.lib.gz_jpeg_scan_end(self.handle, _ptr(n)))
        return int(n[0])

    def jpeg_scan_bits(self):
        """(bits, stuffed bytes) of the last scan: scan bytes = ceil(bits / 8) + stuffed."""
        b = np.zeros(2, np.uint64)
        self._chk(self.L.lib.gz_jpeg_scan_bits(self.handle, _ptr(b[:1]), _ptr(b[1:])))
        return int(b[0]), int(b[1])

    def jpeg_scan_keep(self):
        self._chk(self.L.lib.gz_jpeg_scan_keep(self.handle))

    def jpeg_scan_bytes(self, kept=False, cap=None):
        cap = cap or (self.nb * 3 * 64 * 4 + 1024)
        out = np.zeros(cap, np.uint8)
        n = C.c_size_t(0)
        self._chk(self.L.lib.gz_jpeg_scan_bytes(self.handle, int(kept), _ptr(out), cap,
                                                C.byref(n)))
        return out[:n.value].tobytes()

    # ---- stage probes ----
    def probe_blur(self, plane, sigma, border_ratio):
        p = np.ascontiguousarray(plane, np.float32)
        assert p.shape == (self.h, self.w)
        out = np.zeros_like(p)
        self._chk(self.L.lib.gz_probe_blur(self.handle, _ptr(p), sigma, border_ratio, _ptr(out)))
        return out

    def probe_opsin(self, rgb3):
        p = np.ascontiguousarray(rgb3, np.float32)
        assert p.shape == (3, self.h, self.w)
        out = np.zeros_like(p)
        self._chk(self.L.lib.gz_probe_opsin(self.handle, _ptr(p), _ptr(out)))
        return out

    def probe_separate_frequencies(self, xyb3):
        p = np.ascontiguousarray(xyb3, np.float32)
        out = np.zeros((10, self.h, self.w), np.float32)
        self._chk(self.L.lib.gz_probe_separate_frequencies(self.handle, _ptr(p), _ptr(out)))
        return out

    def probe_diffmap(self, rgb0, rgb1):
        a = np.ascontiguousarray(rgb0, np.float32)
        b = np.ascontiguousarray(rgb1, np.float32)
        d = np.zeros((self.h, self.w), np.float32)
        s = np.zeros(1, np.float32)
        self._chk(self.L.lib.gz_probe_diffmap(self.handle, _ptr(a), _ptr(b), _ptr(d), _ptr(s)))
        return d, float(s[0])

    def probe_mask(self, xyb0, xyb1):
        a = np.ascontiguousarray(xyb0, np.float32)
        b = np.ascontiguousarray(xyb1, np.float32)
        m = np.zeros((3, self.h, self.w), np.float32)
        mdc = np.zeros((3, self.h, self.w), np.float32)
        self._chk(self.L.lib.gz_probe_mask(self.handle, _ptr(a), _ptr(b), _ptr(m), _ptr(mdc)))
        return m, mdc


_default = None


def load():
    """The product library (gfx950).  Raises if it is missing."""
    global _default
    if _default is None:
        # GUETZLI_AMD_LIB: another build of the same library (kernel A/B runs of tools/)
        _default = Library(os.environ.get("GUETZLI_AMD_LIB", DEFAULT_LIB))
    return _default
