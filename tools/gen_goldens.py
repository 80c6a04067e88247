#!/usr/bin/env python3
"""SHA-256 of the UNMODIFIED reference's output (oracle/_ref, built from /root/reference) for
whole encodes that are too slow to run the reference on inside the test-suite.  Run where
/root/reference exists; minutes (small cases) to ~20 minutes (3840x2160) of one core each.  One
script for every family of fixtures under tests/golden/ (it replaces the four per-round scripts):

  gen_goldens.py whole                 -> jpeg_hashes.json         (odd sizes, qualities, synthetic)
  gen_goldens.py jpegin                -> jpeg_input_hashes.json   (JPEG input written by Pillow)
  gen_goldens.py params [--all]        -> params_hashes.json       (guetzli::Params fields, grey, 4:2:0 input)
  gen_goldens.py params_r3 NAME...     -> params_r3/NAME.json      (4:2:0 / q84 / odd size at BASELINE sizes)
  gen_goldens.py config5 K...          -> config5/kK.json          (BASELINE config 5's batch members)
  gen_goldens.py large [NAME...]       -> large/NAME.json          (beyond BASELINE's sizes: 7680x4320,
                                          where an iteration's order exceeds what the device descends
                                          by itself; 75 minutes of one core)
  gen_goldens.py degenerate [NAME...]  -> degenerate/NAME.json (+ NAME.png for PNG input)
                                          (round 4: flat, saturated, noise, slivers, RGBA / 16-bit /
                                          palette PNG through ReadPNG + Process)
  gen_goldens.py photos [NAME...]      -> photos/NAME.json  (round 5: real photographs as RGB and
                                          as camera-written JPEG input, qualities 100 / 99 / 97.5 /
                                          85.5 and the refused 83, a 3840x2160 mosaic without a
                                          period; one file per case, the cases run in parallel
                                          processes: `gen_goldens.py photos --list | xargs -P 6 -n 1
                                          python tools/gen_goldens.py photos`)
"""
import hashlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image
import images
from checkers import ref

GOLD = os.path.join(ROOT, "tests", "golden")
sha = lambda b: hashlib.sha256(b).hexdigest()


def image(spec):
    kind, w, h = spec[:3]
    if kind == "bees":
        return images.bees()
    if kind == "grey":
        return np.repeat(images.tiled(w, h)[:, :, 1:2], 3, axis=2).copy()
    if kind == "flat":
        return images.flat(w, h, spec[3])
    if kind == "stripes":
        return images.stripes(w, h)
    if kind == "noise":
        return images.noise(w, h)
    if kind == "crop":
        return images.crop(w, h, spec[3], spec[4])
    if kind == "photo":
        return images.photo(spec[3])
    if kind == "mosaic":
        return images.mosaic(w, h)
    return images.tiled(w, h) if kind == "tiled" else images.synthetic(w, h)


def pil_jpeg(rgb, pil_kw, tail=b""):
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **pil_kw)
    return b.getvalue() + tail


def pil_json(pil_kw):
    return {k: (v.decode() if isinstance(v, bytes) else v) for k, v in pil_kw.items()}


def params_entry(spec, q, params, pil_kw):
    """One case of guetzli::Process with Params fields; JPEG input when pil_kw is given."""
    rgb = image(spec)
    t0 = time.time()
    target = ref._butteraugli_score_for_quality(q)
    entry = {"image": list(spec), "quality": q, "params": params}
    if pil_kw is None:
        jpg, _ = ref.process_params(rgb, target, **params)
        entry["rgb_sha256"] = sha(rgb.tobytes())
    else:
        data = pil_jpeg(rgb, pil_kw, b"" if params.get("clear_metadata", True) else b"TAIL")
        jpg, _ = ref.process_params(data, target, **params)
        entry["pil"] = pil_json(pil_kw)
        entry["input_sha256"] = sha(data)
    entry["bytes"] = len(jpg)
    entry["jpeg_sha256"] = sha(jpg)
    entry["reference_cpu_seconds"] = round(time.time() - t0, 1)
    return entry


# ------------------------------------------------------------------------------- families --
WHOLE = [("tiled", 1001, 777, 95.0), ("tiled", 612, 408, 88.0), ("synthetic", 640, 480, 95.0),
         ("synthetic", 333, 250, 84.0)]


def fam_whole(args):
    out = {}
    for kind, w, h, q in WHOLE:
        rgb = image((kind, w, h))
        t0 = time.time()
        jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(q))
        key = f"{kind}_{w}x{h}_q{q:g}"
        out[key] = {"rgb_sha256": sha(rgb.tobytes()), "bytes": len(jpg), "jpeg_sha256": sha(jpg)}
        print(key, out[key], f"{time.time() - t0:.0f}s", flush=True)
        json.dump(out, open(os.path.join(GOLD, "jpeg_hashes.json"), "w"), indent=1)


JPEGIN = [("jpegin_612x408_base", 612, 408, dict(quality=96, subsampling=0), 95.0, True),
          ("jpegin_500x333_prog_meta", 500, 333, dict(quality=97, subsampling=0, progressive=True, comment=b"golden"), 90.0, False)]


def fam_jpegin(args):
    """The input stream is written by Pillow at run time; its hash is recorded so that a
    different libjpeg is noticed."""
    out = {}
    for name, w, h, pil_kw, q, clear in JPEGIN:
        data = pil_jpeg(images.tiled(w, h), pil_kw, b"" if clear else b"TAIL")
        t0 = time.time()
        jpg, _ = ref.process_jpeg(data, ref._butteraugli_score_for_quality(q), clear_metadata=clear)
        out[name] = {"kind": "jpeg", "w": w, "h": h, "pil": pil_json(pil_kw), "quality": q,
                     "clear_metadata": clear, "input_sha256": sha(data), "bytes": len(jpg), "jpeg_sha256": sha(jpg)}
        print(name, out[name], f"{time.time() - t0:.0f}s", flush=True)
        json.dump(out, open(os.path.join(GOLD, "jpeg_input_hashes.json"), "w"), indent=1)


PARAMS = [
    # name, image spec, quality, Params fields, Pillow kwargs for JPEG input (or None)
    ("bees_force420_q95", ("bees", 444, 258), 95.0, dict(force_420=True), None),
    ("bees_try420_q90", ("bees", 444, 258), 90.0, dict(try_420=True), None),
    ("tiled_333x251_force420_q84", ("tiled", 333, 251), 84.0, dict(force_420=True), None),
    ("synthetic_320x240_try420_q95", ("synthetic", 320, 240), 95.0, dict(try_420=True), None),
    ("grey_200x150_q95", ("grey", 200, 150), 95.0, dict(), None),
    ("grey_200x150_force420_q90", ("grey", 200, 150), 90.0, dict(force_420=True), None),
    ("tiled_300x200_lookahead2_q95", ("tiled", 300, 200), 95.0, dict(lookahead=2), None),
    ("tiled_300x200_lookahead5_oldmodel_q90", ("tiled", 300, 200), 90.0, dict(lookahead=5, new_model=False), None),
    ("tiled_300x200_lookahead1_q95", ("tiled", 300, 200), 95.0, dict(lookahead=1), None),
    ("jpegin420_612x408", ("tiled", 612, 408), 95.0, dict(), dict(quality=96, subsampling=2)),
    ("jpegin420_501x333_prog_meta", ("tiled", 501, 333), 90.0, dict(clear_metadata=False),
     dict(quality=97, subsampling=2, progressive=True, comment=b"golden")),
    ("jpegin444_try420_400x300", ("tiled", 400, 300), 95.0, dict(try_420=True), dict(quality=97, subsampling=0)),
    ("tiled_333x251_force420_silver_q90", ("tiled", 333, 251), 90.0, dict(force_420=True, silver=True), None),
    ("bees_try420_silver_q95", ("bees", 444, 258), 95.0, dict(try_420=True, silver=True), None),
]


def fam_params(args):
    path = os.path.join(GOLD, "params_hashes.json")
    out = json.load(open(path)) if os.path.exists(path) and "--all" not in args else {}
    for name, spec, q, params, pil_kw in PARAMS:
        if name in out:
            continue
        out[name] = params_entry(spec, q, params, pil_kw)
        out[name].pop("reference_cpu_seconds")
        print(name, out[name], flush=True)
        json.dump(out, open(path, "w"), indent=1)


PARAMS_R3 = {
    "tiled_1920x1080_force420_q95": (("tiled", 1920, 1080), 95.0, dict(force_420=True), None),
    "tiled_1920x1080_try420_q90": (("tiled", 1920, 1080), 90.0, dict(try_420=True), None),
    "tiled_1920x1080_q84": (("tiled", 1920, 1080), 84.0, dict(), None),
    "tiled_1921x1083_q95": (("tiled", 1921, 1083), 95.0, dict(), None),
    "jpegin420_1920x1080": (("tiled", 1920, 1080), 95.0, dict(), dict(quality=96, subsampling=2)),
    "tiled_3840x2160_force420_q95": (("tiled", 3840, 2160), 95.0, dict(force_420=True), None),
}


def fam_params_r3(args):
    """One file per case (the cases run in parallel processes)."""
    os.makedirs(os.path.join(GOLD, "params_r3"), exist_ok=True)
    for name in args:
        entry = params_entry(*PARAMS_R3[name])
        json.dump(entry, open(os.path.join(GOLD, "params_r3", name + ".json"), "w"), indent=1)
        print(name, entry, flush=True)


def fam_config5(args):
    """The 3840x2160 bench image circularly shifted by (37k rows, 53k cols), --quality 95: what
    `xargs -P` does in tests/golden_test.sh:24-26, one process per image."""
    os.makedirs(os.path.join(GOLD, "config5"), exist_ok=True)
    W, H = (int(os.environ.get("C5_W", 3840)), int(os.environ.get("C5_H", 2160)))
    for k in (int(a) for a in args):
        rgb = images.shifted(images.tiled(W, H), k)
        t0 = time.time()
        jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(95.0), cap=W * H * 3 + (1 << 20))
        rec = {"k": k, "w": W, "h": H, "quality": 95.0, "rgb_sha256": sha(rgb.tobytes()),
               "bytes": len(jpg), "jpeg_sha256": sha(jpg), "reference_cpu_seconds": round(time.time() - t0, 1)}
        json.dump(rec, open(os.path.join(GOLD, "config5", f"k{k}.json"), "w"), indent=1)
        print(rec, flush=True)


LARGE = {
    "tiled_7680x4320_q95": (("tiled", 7680, 4320), 95.0, dict(), None),
}


def fam_large(args):
    os.makedirs(os.path.join(GOLD, "large"), exist_ok=True)
    for name in (args or sorted(LARGE)):
        entry = params_entry(*LARGE[name])
        json.dump(entry, open(os.path.join(GOLD, "large", name + ".json"), "w"), indent=1)
        print(name, entry, flush=True)


# Round 4 (VERDICT r3 "content diversity"): content at the edges of what the search handles.
DEGENERATE = {
    # name: (image spec, quality, Params fields)
    "flat_black_256x192_q95": (("flat", 256, 192, (0, 0, 0)), 95.0, dict()),
    "flat_white_255x193_q95": (("flat", 255, 193, (255, 255, 255)), 95.0, dict()),
    "flat_grey_try420_200x136_q90": (("flat", 200, 136, (97, 97, 97)), 90.0, dict(try_420=True)),
    "flat_red_force420_130x70_q84": (("flat", 130, 70, (255, 0, 0)), 84.0, dict(force_420=True)),
    "stripes_320x240_q95": (("stripes", 320, 240), 95.0, dict()),
    "stripes_force420_321x243_q90": (("stripes", 321, 243), 90.0, dict(force_420=True)),
    "noise_512x512_q95": (("noise", 512, 512), 95.0, dict()),
    "noise_try420_160x120_q84": (("noise", 160, 120), 84.0, dict(try_420=True)),
    "sliver_33x700_q95": (("tiled", 33, 700), 95.0, dict()),
    "sliver_700x32_q95": (("tiled", 700, 32), 95.0, dict()),
}



def png_fixtures():
    """PNG inputs the front end converts before Process (guetzli.cc:47-152): RGBA (alpha blended
    on black), 16 bits per sample (high byte), palette + tRNS.  Written with the test-suite's own
    minimal PNG writer, committed as files (a zlib of another version may deflate differently)."""
    import test_png_reader as tp
    b = images.bees()
    h, w = 120, 160
    rgb = b[60:60 + h, 140:140 + w].astype(np.uint32)
    yy, xx = np.mgrid[0:h, 0:w]
    alpha = ((xx * 255) // (w - 1) + (yy * 255) // (h - 1)) // 2
    rgba = np.concatenate([rgb, alpha[..., None].astype(np.uint32)], -1)
    low = (images.noise(w, h, seed=7).astype(np.uint32))
    rgb16 = rgb * 256 + low
    ga16 = np.stack([rgb[..., 1] * 257, (65535 - alpha * 257)], -1).astype(np.uint32)
    pal = images.noise(64, 1, seed=3).reshape(64, 3)
    idx = ((rgb[..., 0] // 64) * 16 + (rgb[..., 1] // 64) * 4 + rgb[..., 2] // 64)[..., None]
    trns = bytes((37 * i) % 256 for i in range(40))
    return {
        "png_rgba8_160x120_q95": (tp.make_png(rgba, 6, 8, filters=(0, 1, 2, 3, 4)), 95.0),
        "png_rgb16_160x120_q95": (tp.make_png(rgb16, 2, 16, filters=(4,)), 95.0),
        "png_greyalpha16_interlaced_160x120_q90": (tp.make_png(ga16, 4, 16, interlace=True, filters=(1, 2)), 90.0),
        "png_palette_trns_160x120_q95": (tp.make_png(idx, 3, 8, palette=pal, trns=trns, filters=(0,)), 95.0),
    }


def fam_degenerate(args):
    out_dir = os.path.join(GOLD, "degenerate")
    os.makedirs(out_dir, exist_ok=True)
    pngs = None
    names = args or (list(DEGENERATE) + ["png_rgba8_160x120_q95", "png_rgb16_160x120_q95",
                                         "png_greyalpha16_interlaced_160x120_q90", "png_palette_trns_160x120_q95"])
    for name in names:
        if name in DEGENERATE:
            spec, q, params = DEGENERATE[name]
            entry = params_entry(spec, q, params, None)
        else:
            import test_png_reader as tp
            pngs = pngs or png_fixtures()
            data, q = pngs[name]
            path = os.path.join(out_dir, name + ".png")
            if os.path.exists(path):
                data = open(path, "rb").read()       # the committed file is the fixture
            else:
                open(path, "wb").write(data)
            rgb = tp.via_reference(data)              # guetzli.cc's ReadPNG over libpng
            t0 = time.time()
            jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(q))
            entry = {"png": name + ".png", "png_sha256": sha(data), "quality": q, "params": {},
                     "rgb_sha256": sha(rgb.tobytes()), "w": int(rgb.shape[1]), "h": int(rgb.shape[0]),
                     "bytes": len(jpg), "jpeg_sha256": sha(jpg), "reference_cpu_seconds": round(time.time() - t0, 1)}
        json.dump(entry, open(os.path.join(out_dir, name + ".json"), "w"), indent=1)
        print(name, entry, flush=True)


# Round 5: photographs and qualities off the four tested so far.
# name: (image spec, quality, Params fields, JPEG-input file or None)
PHOTOS = {
    "china_q95": (("photo", 640, 427, "china"), 95.0, dict(), None),
    "china_q84": (("photo", 640, 427, "china"), 84.0, dict(), None),
    "flower_q95": (("photo", 640, 427, "flower"), 95.0, dict(), None),
    "flower_q84": (("photo", 640, 427, "flower"), 84.0, dict(), None),
    "astronaut_q95": (("photo", 512, 512, "astronaut"), 95.0, dict(), None),
    "astronaut_q84": (("photo", 512, 512, "astronaut"), 84.0, dict(), None),
    "coffee_q95": (("photo", 600, 400, "coffee"), 95.0, dict(), None),
    "coffee_q84": (("photo", 600, 400, "coffee"), 84.0, dict(), None),
    "gravel_q95": (("photo", 512, 512, "gravel"), 95.0, dict(), None),
    "gravel_q84": (("photo", 512, 512, "gravel"), 84.0, dict(), None),
    "chelsea_try420_q90": (("photo", 451, 300, "chelsea"), 90.0, dict(try_420=True), None),
    "rocket_force420_q95": (("photo", 640, 427, "rocket"), 95.0, dict(force_420=True), None),
    "hubble_q92.25": (("photo", 1000, 872, "hubble"), 92.25, dict(), None),
    # the camera's / the editor's own JPEG stream as Process(jpeg_data): its tables, sampling, markers
    "china_jpegin_q95": (None, 95.0, dict(), "china"),
    "flower_jpegin_q95": (None, 95.0, dict(), "flower"),
    "flower_jpegin_keepmeta_q88": (None, 88.0, dict(clear_metadata=False), "flower"),
    "rocket_jpegin_try420_q90": (None, 90.0, dict(try_420=True), "rocket"),
    "retina_jpegin420_q95": (None, 95.0, dict(), "retina"),
    # qualities: the table's end (quality.cc:31-76), the interpolation (:78-85), the refusal
    # (processor.cc:800-806: butteraugli_target > 2.0 -> Process returns false, nothing written)
    "bees_q100": (("bees", 444, 258), 100.0, dict(), None),
    "bees_q99": (("bees", 444, 258), 99.0, dict(), None),
    "bees_q97.5": (("bees", 444, 258), 97.5, dict(), None),
    "bees_q85.5": (("bees", 444, 258), 85.5, dict(), None),
    "bees_q110": (("bees", 444, 258), 110.0, dict(), None),
    "bees_q83_refused": (("bees", 444, 258), 83.0, dict(), None),
    "astronaut_q100": (("photo", 512, 512, "astronaut"), 100.0, dict(), None),
    # BASELINE configs[2]'s size on content without a period
    "mosaic_3840x2160_q95": (("mosaic", 3840, 2160), 95.0, dict(), None),
    "mosaic_1920x1080_q95": (("mosaic", 1920, 1080), 95.0, dict(), None),
    "mosaic_1024x1024_q95": (("mosaic", 1024, 1024), 95.0, dict(), None),           # bench.py's value_1mpix (round 6)
    "mosaic_3840x2160_q84": (("mosaic", 3840, 2160), 84.0, dict(), None),          # BASELINE configs[3]'s quality
    "mosaic_1920x1080_force420_q90": (("mosaic", 1920, 1080), 90.0, dict(force_420=True), None),
    "hubble_jpegin_q99": (None, 99.0, dict(), "hubble"),
}


def fam_photos(args):
    out_dir = os.path.join(GOLD, "photos")
    if args == ["--list"]:
        print("\n".join(PHOTOS))
        return
    for name in (args or list(PHOTOS)):
        spec, q, params, jpeg_in = PHOTOS[name]
        t0 = time.time()
        target = ref._butteraugli_score_for_quality(q)
        entry = {"quality": q, "params": params, "butteraugli_target": target}
        if jpeg_in is None:
            rgb = image(spec)
            entry["image"] = list(spec)
            entry["rgb_sha256"] = sha(rgb.tobytes())
            jpg, _ = ref.process_params(rgb, target, **params)
        else:
            data = images.photo_bytes(jpeg_in)
            entry["jpeg_input"] = images.PHOTO_FILES[jpeg_in]
            entry["input_sha256"] = sha(data)
            jpg, _ = ref.process_params(data, target, **params)
        if jpg is None:
            entry["refused"] = True          # guetzli::Process returned false
        else:
            entry["bytes"] = len(jpg)
            entry["jpeg_sha256"] = sha(jpg)
        entry["reference_cpu_seconds"] = round(time.time() - t0, 1)
        json.dump(entry, open(os.path.join(out_dir, name + ".json"), "w"), indent=1)
        print(name, entry, flush=True)


if __name__ == "__main__":
    fams = {"photos": fam_photos, "whole": fam_whole, "jpegin": fam_jpegin, "params": fam_params, "params_r3": fam_params_r3,
            "config5": fam_config5, "degenerate": fam_degenerate, "large": fam_large}
    if len(sys.argv) < 2 or sys.argv[1] not in fams:
        sys.exit(__doc__)
    fams[sys.argv[1]](sys.argv[2:])
