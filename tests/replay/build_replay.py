"""TEST INFRASTRUCTURE ONLY: builds tests/replay/_build/libgz_replay.so (record/replay shim
of the C ABI, gz_replay.cc) and the product's host driver sources linked against it, so the
host search logic can be exercised at real image sizes without a GPU from a log recorded on
the GPU box.  See gz_replay.cc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libgz_replay.so")
HOST_LIB = os.path.join(OUT, "libguetzli_amd_host_replay.so")


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "gz_replay.cc")
    deps = [src, os.path.join(ROOT, "include", "guetzli_amd.h"),
            os.path.join(ROOT, "guetzli_amd", "csrc", "gz_host_weights.h")]
    if not force and os.path.exists(LIB) and \
            all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", src, "-o", LIB, "-ldl",
                    "-Wall"], check=True)
    return LIB


def build_host(force=False):
    sys.path.insert(0, ROOT)
    from guetzli_amd import build as gzbuild
    return gzbuild.build_host(force=force, device_lib=build(force), out=HOST_LIB)


if __name__ == "__main__":
    print(build_host(force="--force" in sys.argv))
