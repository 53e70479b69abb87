"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/fmx.h declares.
No compute calls here (there is no GPU in this container); the one behavioural check is that the product
refuses to run without a device instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from libfm_amd import build, capi
    build.build()
    return capi.load()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "fmx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fmx_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(lib):
    from libfm_amd import capi
    declared = header_functions()
    bound = sorted(name for name, _, _ in capi.SYMBOLS)
    assert declared == bound
    for name in declared:
        assert hasattr(lib, name), name


def test_exported_symbols_are_c_abi():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "libfm_amd", "libfmx.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name in header_functions():
        assert name in exported, name          # unmangled => extern "C"


def test_abi_version(lib):
    import re
    hdr = open(os.path.join(ROOT, "include", "fmx.h")).read()
    assert lib.fmx_abi_version() == int(re.search(r"#define\s+FMX_ABI_VERSION\s+(\d+)", hdr).group(1)) >= 2


def test_no_cpu_fallback_without_gpu(lib):
    from libfm_amd import capi
    if lib.fmx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.FmxError) as ei:
        capi.Handle(100, 8)
    assert ei.value.code == -2                  # FMX_E_HIP
    assert "no HIP device" in ei.value.text


def test_bad_arguments_are_rejected_before_touching_the_device(lib):
    from libfm_amd import capi
    h = C.c_void_p()
    cfg = capi.Config(0, 8, 1, 1, 0, 0, 0, 0, 0.1, 0, 0, -1, 0, 1, 0)
    assert lib.fmx_create(C.byref(cfg), C.byref(h)) == -1
    assert b"num_attribute" in lib.fmx_last_error(None)
    cfg = capi.Config(10, 8, 1, 1, 7, 0, 0, 0, 0.1, 0, 0, -1, 0, 1, 0)
    assert lib.fmx_create(C.byref(cfg), C.byref(h)) == -1
    assert b"unknown task" in lib.fmx_last_error(None)
    cfg = capi.Config(10, 8, 1, 1, 0, 0, 0, 0, 0.1, 0, 0, -1, 2, 2, 0)
    assert lib.fmx_create(C.byref(cfg), C.byref(h)) == -1
