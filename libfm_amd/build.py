"""Build recipe of the HIP library: libfm_amd/libfmx.so (gfx950 only, in-tree so it travels with gpurun).

    python -m libfm_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "fmx_api.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "fmx_kernels.h"), os.path.join(HERE, "csrc", "fmx_als_kernels.h"),
              os.path.join(ROOT, "include", "fmx.h")]
OUT = os.path.join(HERE, "libfmx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared",
         "-Wno-unused-result", "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + SRC + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
