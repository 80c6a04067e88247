#!/usr/bin/env python3
"""The Compare chain kernel by kernel, and SURVEY.md 8(d)'s block passes, as one JSON that bench.py
puts on its line (`roofline.kernels`, `block_passes`) under the source-digest guard:

    kernel_roofline.py --stats4k CSV --stats1080 CSV [--pmc4k CSV --pmc1080 CSV] [--sq4k CSV --sq1080 CSV] > profiles/rNN_compare_kernels.json

  --stats*  rocprofv3 --kernel-trace --stats summary of tools/run_compare.py with GZ_SINGLE_STREAM=1
            (per-kernel average duration, chain serialised on one stream)
  --pmc*    tools/pmc_summary.py of the FETCH_SIZE / WRITE_SIZE passes (per-kernel averages; KiB;
            FETCH_SIZE counts 128-byte requests as 64 on gfx950: doubled, MI355X_MICROARCH.md)
  --sq*     tools/pmc_summary.py of the SQ passes (SQ_INSTS_VALU, SQ_LDS_BANK_CONFLICT, ...)
Per kernel: launches per Compare, us, counter bytes, TB/s = bytes / us, share of the 8 TB/s HBM peak,
VALU wave-instructions per launch.  Block passes: algorithmic bytes per pixel / us."""
import argparse, csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12
PX = {"4k": 3840 * 2160, "1080p": 1920 * 1080}
BLOCK_PASSES = {   # compulsory bytes per pixel of the integer block passes
    "k_reconstruct": (18.0, "6 B/px of int16 coefficients in, 12 B/px of linear-RGB float planes out (gz_reconstruct: IDCT + colour + sRGB LUT)"),
    "k_quantize": (12.0, "6 B/px in, 6 B/px out (gz_quantize)"),
    "k_encode_rgb": (9.0, "3 B/px of sRGB in, 6 B/px of coefficients out (gz_encode_rgb: RGBToYUV16 + FDCT + /16)"),
}
NOT_IN_CHAIN = ("k_encode_rgb", "k_linear_from_rgb8", "k_quantize", "__amd_rocclr", "k_mask_sup")


def norm(name):
    name = re.sub(r"^void ", "", name.strip().strip('"'))
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"gz::", "", name)


def rows(path):
    return list(csv.DictReader(l for l in open(path) if not l.startswith("#")))


def head_of(path):
    first = open(path).readline()
    m = re.match(r"#\s*head\s+(\S+)", first)
    return m.group(1) if m else None


def per_size(size, stats, pmc, sq):
    st = {norm(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in rows(stats)}
    compares = st["k_combine"][0]
    ctr = {}
    for path in (pmc, sq):
        if path:
            for r in rows(path):
                ctr.setdefault(norm(r["kernel"]), {})[r["counter"]] = float(r["avg_value"])
    out, total_us, total_b = [], 0.0, 0.0
    for k, (calls, us) in sorted(st.items(), key=lambda kv: -kv[1][1] * max(1, kv[1][0] // compares)):
        if any(x in k for x in NOT_IN_CHAIN):
            continue
        per = max(1, calls // compares)
        c = ctr.get(k, {})
        e = {"kernel": k, "launches_per_compare": per, "us": round(us, 1)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            b = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            e.update(counter_bytes=round(b), tbps=round(b / us / 1e6, 2), frac_of_hbm_peak=round(b / (us * 1e-6) / HBM_PEAK, 3))
            total_b += per * b
        if "SQ_INSTS_VALU" in c:
            e["valu_wave_instructions"] = round(c["SQ_INSTS_VALU"])
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        total_us += per * us
        out.append(e)
    summary = {"compares_in_trace": compares, "launches_per_compare": sum(e["launches_per_compare"] for e in out),
               "sum_us_serialised": round(total_us, 1), "counter_bytes": round(total_b),
               "counter_bytes_per_px": round(total_b / PX[size], 1) if total_b else None}
    blocks = {}
    for k, (bpp, what) in BLOCK_PASSES.items():
        if k in st:
            us = st[k][1]
            blocks[k] = {"algorithmic_bytes_per_px": bpp, "what": what, "us": round(us, 1),
                         "achieved_gbps": round(bpp * PX[size] / us / 1e3, 1),
                         "frac": round(bpp * PX[size] / (us * 1e-6) / HBM_PEAK, 3), "workload": size}
    return {"kernels": out, "summary": summary}, blocks


def main():
    ap = argparse.ArgumentParser()
    for n in ("stats4k", "stats1080", "pmc4k", "pmc1080", "sq4k", "sq1080"):
        ap.add_argument("--" + n)
    a = ap.parse_args()
    from guetzli_amd.build import csrc_digest
    res = {"note": "per kernel of one butteraugli Compare: average duration with the chain serialised on one "
                   "stream (rocprofv3 --kernel-trace --stats, GZ_SINGLE_STREAM=1), HBM bytes from the "
                   "FETCH_SIZE / WRITE_SIZE passes (2 x FETCH + WRITE), VALU wave-instructions per launch; "
                   "tools/kernel_roofline.py",
           "csrc_sha256": csrc_digest(), "head": head_of(a.stats4k) if a.stats4k else None}
    blocks = {}
    if a.stats4k:
        res["4k"], b = per_size("4k", a.stats4k, a.pmc4k, a.sq4k)
        res["4k"], blocks = res["4k"], b
        res["4k"] = {"workload": "3840x2160", **res["4k"]}
    if a.stats1080:
        r, b = per_size("1080p", a.stats1080, a.pmc1080, a.sq1080)
        res["1080p"] = {"workload": "1920x1080", **r}
        if not blocks:
            blocks = b
    res["block_passes"] = blocks
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
