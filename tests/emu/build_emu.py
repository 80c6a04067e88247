"""TEST INFRASTRUCTURE ONLY: compiles the product's kernel sources (guetzli_amd/csrc) with
g++ against tests/emu/hip_emu.h into tests/emu/_build/libguetzli_amd_emu.so, so that the
CPU test-suite can check the kernels' indexing and arithmetic order against the oracle
without a GPU.  See hip_emu.h for why this is not a CPU fallback of the product."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "guetzli_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libguetzli_amd_emu.so")


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(d, f) for d, _, fs in os.walk(CSRC) for f in fs] + \
        [os.path.join(HERE, "hip_emu.h"), os.path.join(ROOT, "include", "guetzli_amd.h")]
    first = [True]

    def stale():
        f, first[0] = first[0], False
        return (force and f) or not os.path.exists(LIB) or \
            any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)
    if not stale():
        return LIB
    first[0] = True
    import sys
    sys.path.insert(0, ROOT)
    from guetzli_amd import build as gzbuild
    # (one of the processes pytest -n starts together compiles, the others wait and load the finished library)
    return gzbuild.locked_compile(LIB, stale, lambda tmp: [
        "g++", "-O2", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DGZ_EMU",
        "-I" + HERE, "-x", "c++", os.path.join(CSRC, "gz_api.hip"), "-o", tmp,
        "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-unused-variable"])


HOST_LIB = os.path.join(OUT, "libguetzli_amd_host_emu.so")


def build_host(force=False):
    """The product's host driver sources linked against the EMULATED device library, so
    that a whole encode can be checked on the CPU against the reference."""
    import sys
    sys.path.insert(0, ROOT)
    from guetzli_amd import build as gzbuild
    return gzbuild.build_host(force=force, device_lib=build(), out=HOST_LIB)


if __name__ == "__main__":
    print(build(force=True))
