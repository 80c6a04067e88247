"""Builds the product library guetzli_amd/libguetzli_amd.so for gfx950 with hipcc.

In-tree, explicit hipcc (no JIT cache): the built .so travels to the GPU box with the
repository snapshot.  Flags that matter for result parity (SURVEY.md §7 "hard parts"):
  -ffp-contract=off                 no FMA contraction (the reference is SSE2, no FMA)
  no -ffast-math, no -fgpu-flush-denormals-to-zero; correctly rounded f32 divide/sqrt is
  the HIP default (-fhip-fp32-correctly-rounded-divide-sqrt).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libguetzli_amd.so")
SOURCES = ["gz_api.hip"]
HEADERS = ["gz_common.h", "gz_math.h", "gz_kernels_block.h", "gz_kernels_blur.h",
           "gz_kernels_diff.h", "gz_kernels_search.h", "gz_kernels_entropy.h", "gz_kernels_dctd.h", "gz_kernels_downsample.h",
           "gz_kernels_order.h", "gz_kernels_rank.h", "gz_host_weights.h", "tables_generated.h", "order_tables_generated.h",
           # gz_api.hip by concern (one translation unit)
           "api/plans.h", "api/context.h", "api/chain.h", "api/entry_context.h", "api/entry_compare.h",
           "api/entry_phaseb.h", "api/entry_entropy.h", "api/entry_frame.h", "api/entry_search.h", "api/entry_probes.h"]
ARCH = "gfx950"
# -vectorize-slp=false: left to itself the compiler packs adjacent scalar f32 adds / multiplies
# into v_pk_*_f32 pairs; those issue at the same lane rate as the scalar forms on gfx950
# (tools/ubench/pk.hip) but need their operands in aligned register pairs (extra moves) and a
# wait state before a dependent use: Malta 323 -> 304 us, the whole chain -3 % (4K) / -5 %
# (1080p) with it off (profiles/r02_packed_blur_and_malta_diff_experiments.log, section 8).  The
# kernels that want packed arithmetic ask for it explicitly (gz_f2).  Same results either way.
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-mllvm", "-vectorize-slp=false",
         # the fused blurs' taps in vector registers (gz_common.h GZ_IN_VGPR: a scalar-register operand
         # makes v_mul_f32 a 4-cycle instruction, tools/ubench/issue.hip): chain -0.7 % at 4K beside
         # Malta, nothing serialised (profiles/r05_variants.log)
         "-DGZ_TAPS_VGPR"]


def _newer_than_lib():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(os.path.dirname(HERE), "include", "guetzli_amd.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# What decides the kernels' work and traffic: the kernels, their arithmetic and tables, and the chain's
# dispatch (which instantiation / tile shape for which size).  The rest of the C ABI's host side
# (contexts, pools, entry points) changes nothing a profile of the chain measures.
DIGEST_FILES = [h for h in HEADERS if not h.startswith("api/")] + \
    ["api/plans.h", "api/chain.h", "api/entry_search.h", "api/entry_phaseb.h", "api/entry_entropy.h"]   # (launch geometry lives there too: ADVICE r5)


def csrc_digest():
    """SHA-256 over the kernel sources and the chain's dispatch (DIGEST_FILES, in name order): what a
    profile under profiles/ was measured on.  bench.py refuses a committed traffic figure whose digest is
    not the tree's (the GPU box's snapshot has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(DIGEST_FILES):
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def locked_compile(out, stale, make_cmd, verbose=False):
    """One process compiles `out`, the others wait for it: under an exclusive lock on <out>.lock the staleness test
    `stale()` is repeated, the compiler writes <out>.tmp.<pid> and the result is renamed into place -- so that
    processes started together (pytest -n 4 after a source change) neither compile the same file four times nor
    load a library another one is still writing.  make_cmd(tmp_path) -> argv."""
    import fcntl
    with open(out + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not stale():
            return out
        tmp = f"{out}.tmp.{os.getpid()}"
        cmd = make_cmd(tmp)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        try:
            subprocess.run(cmd, check=True)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    return out


def build(force=False, verbose=False):
    """Compile the HIP extension if sources are newer than the library."""
    if not force and not _newer_than_lib():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        if os.path.exists(LIB):
            return LIB     # prebuilt library shipped with the snapshot, no compiler here
        raise RuntimeError("hipcc not found and no prebuilt libguetzli_amd.so")
    # One code object, gfx950 with XNACK off: code compiled for a known XNACK mode schedules its memory
    # clauses more freely than the "any" default -- the Compare chain runs 2-4 % (1080p) / 1.6 % (4K)
    # faster (profiles/r02_packed_blur_and_malta_diff_experiments.log, section 9).  Until round 5 an
    # xnack+ code object rode along for processes started with HSA_XNACK=1; this pool runs XNACK off only
    # (and refuses libraries that carry xnack+ code), and that is what MI355X boxes default to.
    first = [True]

    def stale():   # (the caller's `force` counts once: whoever waited for the lock finds the library fresh)
        f, first[0] = first[0], False
        return (force and f) or _newer_than_lib()
    return locked_compile(LIB, stale, lambda tmp: [hipcc, f"--offload-arch={ARCH}:xnack-"] + FLAGS +
                          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp], verbose)


def build_variant(name, defines):
    """A/B builds for tools/gpu_variants.sh: guetzli_amd/variants/<name>.so = the library compiled
    with extra -D definitions (build-time experiments of the kernels; git-ignored, travels to the GPU
    box, where the script copies one variant after the other over libguetzli_amd.so)."""
    out = os.path.join(HERE, "variants", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc_path()] + [f"--offload-arch={ARCH}:xnack-"] + FLAGS + ["-D" + d for d in defines] + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    _run(cmd, True)
    return out


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libguetzli_amd_host.so")
HOST_SOURCES = ["jpeg_reader.cc", "jpeg_writer.cc", "png_reader.cc", "processor.cc", "silver_screen.cc"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra",
              "-Wno-unused-parameter"]


def build_host(force=False, verbose=False, device_lib=None, out=None):
    """The host search driver (C++, g++) linked against the C-ABI library."""
    device_lib = device_lib or LIB
    out = out or HOST_LIB
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    # every header of the directory (round 5's code_refresh.h was missing from a hand-kept list: VERDICT r5)
    deps = srcs + sorted(os.path.join(HOST_DIR, h) for h in os.listdir(HOST_DIR) if h.endswith(".h")) + \
        [os.path.join(os.path.dirname(HERE), "include", "guetzli_amd.h"), device_lib]
    if not force and os.path.exists(out) and \
            all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps if os.path.exists(d)):
        return out
    if shutil.which("g++") is None:
        if os.path.exists(out):
            return out
        raise RuntimeError("g++ not found and no prebuilt host library")
    libdir, libname = os.path.split(device_lib)
    first = [True]

    def stale():
        f, first[0] = first[0], False
        return (force and f) or not os.path.exists(out) or \
            any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps if os.path.exists(d))
    return locked_compile(out, stale, lambda tmp: ["g++"] + HOST_FLAGS + srcs +
                          ["-o", tmp, "-L" + libdir, "-l:" + libname, "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,$ORIGIN", "-lz"], verbose)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":   # build.py --variant NAME [DEFINE ...]
        print(build_variant(sys.argv[2], sys.argv[3:]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
