#!/usr/bin/env python3
"""HBM traffic of one butteraugli Compare chain from the per-kernel PMC summaries that
tools/gpu_pmc.sh + tools/pmc_summary.py produce (FETCH_SIZE / WRITE_SIZE, separate passes).
FETCH_SIZE is doubled: gfx950 counts 128-byte requests as 64 bytes (calibrated on the plane
copy of tools/ubench/bw in the same session: bw_pmc.csv).  Compare chains per run: 3.
Usage: pmc_traffic_json.py compare_4k_pmc.csv compare_1080p_pmc.csv bw_pmc.csv > traffic.json"""
import csv, json, sys

CHAIN = ("k_blur", "k_malta", "k_mask_pre", "k_combine", "k_reconstruct")


def chain_totals(path, compares):
    """Per chain: every kernel's average counter value x its launches per chain (its calls in
    the run / the run's Compare count, rounded: the context set-up adds one extra launch of
    the kernels that also build the original's PsychoImage)."""
    fetch = write = 0.0
    for r in csv.DictReader(open(path)):
        if not any(k in r["kernel"] for k in CHAIN):
            continue
        per_chain = round(int(r["calls"]) / compares)   # (0: a kernel of the context set-up only)
        total = float(r["avg_value"]) * per_chain
        if r["counter"] == "FETCH_SIZE":
            fetch += total
        elif r["counter"] == "WRITE_SIZE":
            write += total
    return fetch, write


def repo_head():
    """The commit the measured code was built from: `git rev-parse HEAD`, or -- on the GPU box, whose
    snapshot has no .git -- the file tools/gpurun_head.sh wrote before the run."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        h = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True)
        if h.returncode == 0 and h.stdout.strip():
            return h.stdout.strip()
    except Exception:
        pass
    try:
        return open(os.path.join(root, ".gpurun_head")).read().strip()
    except Exception:
        return None


def main():
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_pmc.sh) of one "
                   "butteraugli Compare chain (the chain's kernels: per-kernel average x launches per chain); "
                   "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, see the "
                   "calibration entry); traffic = (2*FETCH+WRITE)*1024"}
    # run_compare.py W H 3 runs 3 timed chains + the set-up's chains; count them from k_malta
    for key, path, px in (("4k", sys.argv[1], 3840 * 2160), ("1080p", sys.argv[2], 1920 * 1080)):
        n = 0
        for r in csv.DictReader(open(path)):
            if "k_malta" in r["kernel"] and r["counter"] == "FETCH_SIZE":
                n = int(r["calls"])
        f, w = chain_totals(path, n)
        out[key] = {"compares_in_run": n, "fetch_size_kb_raw": round(f, 1), "write_size_kb": round(w, 1),
                    "traffic_bytes": (2 * f + w) * 1024, "bytes_per_px": (2 * f + w) * 1024 / px}
    cal = {}
    for r in csv.DictReader(open(sys.argv[3])):
        if "dwordx4" in r["kernel"] or "copy" in r["kernel"].lower():
            cal.setdefault(r["kernel"], {})[r["counter"]] = float(r["avg_value"])
    out["calibration"] = cal
    out["head"] = repo_head()
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from guetzli_amd.build import csrc_digest
    out["csrc_sha256"] = csrc_digest()      # bench.py: a figure measured on other kernel sources is stale
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
