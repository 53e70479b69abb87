"""Build recipe of the HIP library: libfm_amd/libfmx.so (gfx950 only, in-tree so it travels with gpurun).

    python -m libfm_amd.build [--force]

Five translation units (csrc/fmx_core.hip, fmx_sgd.hip, fmx_als.hip, fmx_io.hip, fmx_comm.hip) are compiled in parallel and linked.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SRC = [os.path.join(CSRC, f) for f in ("fmx_core.hip", "fmx_sgd.hip", "fmx_als.hip", "fmx_io.hip", "fmx_comm.hip")]
DEPS = SRC + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "fmx.h")]   # every header under csrc/
# experiments: FMX_DEFS="-DFMX_V_NT=1" FMX_OUT=libfmx_nt.so python -m libfm_amd.build --force ; run with FMX_LIB=<that file>
EXTRA = os.environ.get("FMX_DEFS", "").split()
OUT = os.path.join(HERE, os.environ.get("FMX_OUT", "libfmx.so"))
OBJ_DIR = os.path.join(HERE, "build" + ("_" + os.environ["FMX_OUT"].replace(".", "_") if "FMX_OUT" in os.environ else ""))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-fvisibility=hidden",
         "-Wno-unused-result", "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = [os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in SRC]

    def cc(pair):
        cmd = [HIPCC] + FLAGS + EXTRA + ["-c", pair[0], "-o", pair[1]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(len(SRC)) as ex:
        list(ex.map(cc, zip(SRC, objs)))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
