#!/usr/bin/env python3
"""Encode times with the package under PKGDIR -- A/B of two BUILDS on one box (boxes differ by +-4 %, processes on
one box by less).  The other build: `git archive <commit> guetzli_amd include | tar -x -C /tmp/old`, build it there
(python -c "import sys; sys.path.insert(0, '/tmp/old'); from guetzli_amd import build; build.build(); build.build_host()"),
`cp -a /tmp/old/guetzli_amd /tmp/old/include tools/ab_old/` (git-ignored; built libraries travel with gpurun; the host
library finds the device library through $ORIGIN), then on the GPU box in turn:
    python tools/ab_old_time.py tools/ab_old W H REPS;  python tools/ab_old_time.py . W H REPS
Usage: ab_old_time.py PKGDIR W H REPS   (PKGDIR holds guetzli_amd/)"""
import sys, os, time
pkg, w, h, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.abspath(pkg))
import guetzli_amd, images
rgb = images.tiled(w, h)
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); jpg, info = guetzli_amd.process(rgb, quality=95.0); ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
it = info["counters"]["number of iterations"]
print(f"{os.path.abspath(pkg)}: {w}x{h} median {ts[len(ts)//2]*1e3:.2f} ms, min {ts[0]*1e3:.2f}; {it} iterations -> {ts[len(ts)//2]*1e6/it:.1f} us per iteration ({guetzli_amd.__file__})")
