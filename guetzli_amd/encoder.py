"""Python binding of the host search driver (guetzli_amd/host, libguetzli_amd_host.so):
guetzli_amd.process(rgb, quality=95) == guetzli::Process(params, stats, rgb, w, h, &out)
with the numeric hot path on the MI355X.  No CPU fallback: the host library links the
gfx950 C-ABI library and fails if no GPU is usable."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_HOST_LIB = os.path.join(HERE, "libguetzli_amd_host.so")


def _jpeg_dimensions(data):
    """(width, height) of the first SOFn segment of a JPEG stream, (0, 0) if there is none."""
    i, n = 2, len(data)
    while i + 9 < n:
        if data[i] != 0xFF:
            i += 1
            continue
        m = data[i + 1]
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7 or m == 0xFF:
            i += 2 if m != 0xFF else 1
            continue
        seg = (data[i + 2] << 8) | data[i + 3]
        if 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            return (data[i + 7] << 8) | data[i + 8], (data[i + 5] << 8) | data[i + 6]
        i += 2 + seg
    return 0, 0


class HostLibrary:
    def __init__(self, path=DEFAULT_HOST_LIB):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: run `python -m guetzli_amd.build`")
        self.lib = C.CDLL(path)
        self.lib.gzh_process.restype = C.c_long
        self.lib.gzh_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_float,
                                         C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                         C.c_void_p, C.c_long]
        self.lib.gzh_process_jpeg.restype = C.c_long
        self.lib.gzh_process_jpeg.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_float, C.c_int,
                                              C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        self.lib.gzh_process_params.restype = C.c_long
        self.lib.gzh_process_params.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_double,
                                                C.c_float, C.c_void_p, C.c_void_p, C.c_long,
                                                C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        self.lib.gzh_write_jpeg_factor.restype = C.c_long
        self.lib.gzh_write_jpeg_factor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_int, C.c_void_p, C.c_long]
        self.lib.gzh_jpeg_head_factor.restype = C.c_long
        self.lib.gzh_jpeg_head_factor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        self.lib.gzh_read_png.restype = C.c_long
        self.lib.gzh_read_png.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long]
        self.lib.gzh_write_jpeg.restype = C.c_long
        self.lib.gzh_write_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_long]
        self.lib.gzh_jpeg_head.restype = C.c_long
        self.lib.gzh_jpeg_head.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        self.lib.gzh_butteraugli_score_for_quality.restype = C.c_double
        self.lib.gzh_butteraugli_score_for_quality.argtypes = [C.c_double]

    def _process(self, data, w, h, quality, target, device, clear_metadata, try_420, force_420,
                 use_silver_screen, lookahead, new_model, want_trace):
        """gzh_process_params with a buffer that grows to what the library asks for."""
        is_jpeg = isinstance(data, (bytes, bytearray))
        buf = np.frombuffer(data, np.uint8) if is_jpeg else data
        if is_jpeg:
            # from the frame header's dimensions (a heavily compressed input re-encoded at a high
            # quality can be many times its own size; a second call would repeat the whole search)
            jw, jh = _jpeg_dimensions(data)
            if jw * jh > (1 << 28):   # a header nobody has validated yet (65535 x 65535 = 12.9 GB):
                jw = jh = 0           # start small, the grow-and-retry path covers a real giant
            cap = max(3 * jw * jh + (1 << 16), 4 * len(data), 1 << 20)
        else:
            cap = 3 * w * h + (1 << 16)
        ip = (C.c_int * 7)(device, int(clear_metadata), int(try_420), int(force_420),
                           int(use_silver_screen), int(lookahead), int(new_model))
        tr = C.create_string_buffer(1 << 24) if want_trace else None
        tm = C.create_string_buffer(1 << 12)
        for _ in range(2):
            out = np.empty(cap, np.uint8)
            n = self.lib.gzh_process_params(buf.ctypes.data, len(data) if is_jpeg else -1, w, h,
                                            -1.0 if target is not None else float(quality),
                                            float(target or 0.0), ip, out.ctypes.data, cap,
                                            tr, len(tr) if tr else 0, tm, len(tm))
            if n < 0:
                raise RuntimeError("guetzli_amd.Process failed (see stderr)" if n == -1 else
                                   "guetzli_amd.Process raised a C++ exception (see stderr)")
            if n <= cap:
                break
            cap = n   # the JPEG did not fit (nothing was copied): once more with room for it
        else:
            raise RuntimeError(f"guetzli_amd.Process: output of {n} bytes does not fit")
        timers, counters = {}, {}
        for item in tm.value.decode().split(";"):
            if "=" in item:
                k, v = item.split("=")
                if k.startswith("#"):
                    counters[k[1:]] = int(v)
                else:
                    timers[k] = float(v)
        return out[:n].tobytes(), {"trace": tr.value.decode() if tr else None,
                                   "timers": timers, "counters": counters}

    def process(self, rgb, quality=95.0, target=None, device=0, want_trace=False, try_420=False,
                force_420=False, use_silver_screen=False, lookahead=3, new_model=True):
        """guetzli::Process(params, stats, rgb, w, h, &out).  Returns (jpeg_bytes, info) where
        info has 'trace' (the --verbose text, if requested), 'timers' (seconds per phase) and
        'counters'.  The keyword arguments are the fields of guetzli::Params."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, ch = rgb.shape
        assert ch == 3
        return self._process(rgb, w, h, quality, target, device, True, try_420, force_420,
                             use_silver_screen, lookahead, new_model, want_trace)

    def read_png(self, data):
        """ReadPNG of the reference's front end (guetzli.cc:47-152): PNG bytes -> uint8
        [h][w][3] with alpha blended on black.  Raises if the stream is rejected."""
        buf = np.frombuffer(data, np.uint8)
        wh = (C.c_int * 2)()
        n = self.lib.gzh_read_png(buf.ctypes.data, len(data), wh, None, 0)
        if n < 0:
            raise ValueError("not a readable PNG (see stderr)")
        out = np.zeros(n, np.uint8)
        n2 = self.lib.gzh_read_png(buf.ctypes.data, len(data), wh, out.ctypes.data, n)
        assert n2 == n
        return out.reshape(wh[1], wh[0], 3)

    def process_jpeg(self, data, quality=95.0, target=None, device=0, clear_metadata=True,
                     want_trace=False, try_420=False, force_420=False, use_silver_screen=False,
                     lookahead=3, new_model=True):
        """guetzli::Process(params, stats, jpeg_data, &out) for a YUV 4:4:4 or 4:2:0 JPEG.
        Returns (jpeg_bytes, trace) or raises if the input is refused (message on stderr)."""
        jpg, info = self._process(bytes(data), 0, 0, quality, target, device, clear_metadata,
                                  try_420, force_420, use_silver_screen, lookahead, new_model,
                                  want_trace)
        return jpg, info["trace"]

    def write_jpeg(self, coeffs, w, h, q=None, factor=1):
        """WriteJpeg of dequantised coefficients (frame layout of include/guetzli_amd.h:
        [3][nb][64] for factor 1, nb + 2*nbc blocks for the 4:2:0 factor 2) with quant matrices
        q[3][64]; q=None writes the q=1 'original' frame of EncodeRGBToJpeg."""
        co = np.ascontiguousarray(coeffs, np.int16)
        qq = None if q is None else np.ascontiguousarray(q, np.int32)
        cap = 6 * w * h + (1 << 16)
        for _ in range(2):
            out = np.zeros(cap, np.uint8)
            n = self.lib.gzh_write_jpeg_factor(co.ctypes.data, w, h,
                                               qq.ctypes.data if qq is not None else None,
                                               int(q is None), factor, out.ctypes.data, cap)
            if n < 0:
                raise RuntimeError("WriteJpeg failed")
            if n <= cap:
                return out[:n].tobytes()
            cap = n
        raise RuntimeError("WriteJpeg: output does not fit")

    def jpeg_head(self, counts, w, h, q=None, ncomp=3, factor=1):
        """SOI..SOS bytes + per-component Huffman codes (depth, code: [2][3][256]) from the
        symbol counts of gz_jpeg_histograms; q=None is the q=1 'original' frame."""
        cnt = np.ascontiguousarray(counts, np.uint32)
        assert cnt.shape == (2, 3, 256)
        qq = None if q is None else np.ascontiguousarray(q, np.int32)
        head = np.zeros(1 << 16, np.uint8)
        depth = np.zeros((2, 3, 256), np.uint8)
        code = np.zeros((2, 3, 256), np.uint16)
        n = self.lib.gzh_jpeg_head_factor(cnt.ctypes.data, qq.ctypes.data if qq is not None else None,
                                          w, h, ncomp, factor, head.ctypes.data, head.size,
                                          depth.ctypes.data, code.ctypes.data)
        assert 0 <= n <= head.size, n
        return head[:n].tobytes(), depth, code

    def butteraugli_score_for_quality(self, q):
        return self.lib.gzh_butteraugli_score_for_quality(q)


_default = None


def load_host():
    global _default
    if _default is None:
        _default = HostLibrary()
    return _default


def process(rgb, quality=95.0, **kw):
    return load_host().process(rgb, quality=quality, **kw)


def read_png(data):
    """PNG bytes -> uint8 [h][w][3], as the reference's front end reads them (alpha on black)."""
    return load_host().read_png(data)


def process_png(data, quality=95.0, **kw):
    """`guetzli --quality Q in.png out.jpg`: ReadPNG + Process.  Returns (jpeg_bytes, info)."""
    return load_host().process(load_host().read_png(data), quality=quality, **kw)
