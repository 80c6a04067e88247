#!/usr/bin/env python3
"""Which unit is busy in batch mode.  Sums rocprofv3 --pmc counters of a whole
batch run per kernel (not per-launch averages: the batch's images differ in launch counts) and per
image, and prices the sums against the chip for a given GPU time per image:

    batch_pmc_summary.py DIR [DIR...] [--images N] [--seconds-per-image S] [--clock-ghz F]

stdout: CSV kernel,launches,<counter sums...>;  stderr: the per-image totals and busy fractions.
Units (MI355X_MICROARCH.md, rocprofv3 PMC): SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count
quad-cycles per wavefront summed over the chip; SQ_LDS_IDX_ACTIVE LDS-array cycles summed over the
CUs; FETCH_SIZE / WRITE_SIZE KiB (FETCH_SIZE tallies 128-byte requests as 64: doubled here).  The
profiler serialises the dispatches, so the durations of such a run mean nothing -- the sums are what
the co-running batch has to execute, and S (measured WITHOUT the profiler) is what it takes."""
import csv, glob, os, re, sys
from collections import defaultdict

args = sys.argv[1:]
opt = {"--images": 4.0, "--seconds-per-image": 0.22, "--clock-ghz": 2.4}
dirs = []
i = 0
while i < len(args):
    if args[i] in opt:
        opt[args[i]] = float(args[i + 1]); i += 2
    else:
        dirs.append(args[i]); i += 1
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"]))
            name = re.sub(r"^gz::", "", name)
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name][row["Counter_Name"]] += 1
ctrs = sorted({c for k in tot for c in tot[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches"] + ctrs)
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_ACTIVE_INST_VALU", 0)):
    w.writerow([k, max(calls[k].values())] + [f"{tot[k].get(c, 0):.0f}" for c in ctrs])

n_img, sec, ghz = opt["--images"], opt["--seconds-per-image"], opt["--clock-ghz"]
S = lambda c: sum(tot[k].get(c, 0) for k in tot)
cyc = sec * ghz * 1e9                      # cycles one image may take
e = sys.stderr
print(f"# batch of {n_img:g} images; priced at {sec} s of GPU per image, {ghz} GHz", file=e)
if S("SQ_ACTIVE_INST_VALU"):
    valu = S("SQ_ACTIVE_INST_VALU") * 4 / n_img / 1024          # cycles per SIMD and image
    print(f"VALU:  {S('SQ_INSTS_VALU') / n_img / 1e9:.2f} G wave-instructions per image, "
          f"{valu / 1e6:.1f} M busy cycles per SIMD = {valu / cyc:.2f} of the time "
          f"(quad-cycle counter: an upper bound where instructions issue in 2 cycles)", file=e)
    lds_i = S("SQ_ACTIVE_INST_LDS") * 4 / n_img / 1024
    print(f"LDS instructions: {S('SQ_INSTS_LDS') / n_img / 1e9:.2f} G per image, issue-active {lds_i / cyc:.2f} of the time per SIMD", file=e)
    occ = S("SQ_WAVE_CYCLES") * 4 / n_img / (256 * 32)
    print(f"resident wavefronts: {S('SQ_WAVE_CYCLES') * 4 / n_img / 1e9:.1f} G wave-cycles per image "
          f"= {occ / cyc:.2f} of the chip's 8192 wave slots for that time "
          f"({32 * occ / cyc:.1f} of 32 per CU)", file=e)
    print(f"wave cycles waiting to issue (SQ_WAIT_INST_ANY): {S('SQ_WAIT_INST_ANY') / max(S('SQ_WAVE_CYCLES'), 1):.2f}", file=e)
if S("SQ_LDS_IDX_ACTIVE"):
    lds = S("SQ_LDS_IDX_ACTIVE") / n_img / 256
    print(f"LDS array: {lds / 1e6:.1f} M active cycles per CU and image = {lds / cyc:.2f} of the time; "
          f"bank conflicts {S('SQ_LDS_BANK_CONFLICT') / max(S('SQ_LDS_IDX_ACTIVE'), 1):.2f} of them", file=e)
    print(f"wave cycles parked (SQ_WAIT_ANY): {S('SQ_WAIT_ANY') / max(S('SQ_WAVE_CYCLES'), 1) if S('SQ_WAVE_CYCLES') else float('nan'):.2f}"
          f"  (waves launched per image: {S('SQ_WAVES') / n_img / 1e6:.2f} M)", file=e)
if S("FETCH_SIZE") or S("WRITE_SIZE"):
    rd, wr = 2 * S("FETCH_SIZE") * 1024 / n_img, S("WRITE_SIZE") * 1024 / n_img
    print(f"HBM:   {rd / 1e9:.1f} GB read (FETCH_SIZE x 2) + {wr / 1e9:.1f} GB written per image = "
          f"{(rd + wr) / sec / 1e12:.2f} TB/s over {sec} s = {(rd + wr) / sec / 8e12:.2f} of 8 TB/s, "
          f"{(rd + wr) / sec / 6.3e12:.2f} of the 6.3 TB/s a copy reaches", file=e)
print("# top kernels by VALU busy cycles (share of the batch's VALU / of its LDS array cycles / of its HBM bytes)", file=e)
tv, tl = max(S("SQ_ACTIVE_INST_VALU"), 1), max(S("SQ_LDS_IDX_ACTIVE"), 1)
tb = max(2 * S("FETCH_SIZE") + S("WRITE_SIZE"), 1)
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_ACTIVE_INST_VALU", 0))[:16]:
    t = tot[k]
    print(f"  {k[:58]:58s} {t.get('SQ_ACTIVE_INST_VALU', 0) / tv:5.2f} {t.get('SQ_LDS_IDX_ACTIVE', 0) / tl:5.2f} "
          f"{(2 * t.get('FETCH_SIZE', 0) + t.get('WRITE_SIZE', 0)) / tb:5.2f}", file=e)
