"""The C-ABI used from plain C (examples/fmx_demo.c): links against libfmx.so with gcc, no Python/torch in the process.

Without a GPU the program must fail loudly (no CPU fallback); on the GPU it must run the three SGD modes and ALS."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "fmx_demo")


def _build():
    lib = os.path.join(ROOT, "libfm_amd", "libfmx.so")
    if not os.path.exists(lib):
        pytest.skip("libfmx.so not built")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "fmx_demo.c"),
           "-L" + os.path.join(ROOT, "libfm_amd"), "-lfmx", "-Wl,-rpath," + os.path.join(ROOT, "libfm_amd"),
           "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_c_demo_fails_loudly_without_gpu():
    if _has_gpu():
        pytest.skip("a GPU is present; covered by the gpu test")
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_demo_runs_on_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
    for name in ("sequential", "minibatch", "hogwild", "als"):
        assert name in r.stdout
