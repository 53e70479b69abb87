"""GPU: SURVEY section 8 row f2 end to end -- the REFERENCE's own tools write the binary files (tools/convert.cpp, tools/transpose.cpp;
LargeSparseMatrixHD format, src/util/fmatrix.h:44-50,165-230), fmx_read_binary lands them in page-locked buffers (Data::load's binary
branch, src/libfm/src/Data.h:119-178), fmx_upload_rows takes exactly those buffers, and the learners run on them: one SGD epoch
and one ALS sweep against the oracle on the same rows.  Also: the upload from the page-locked buffers is not slower than from
pageable memory (it is a DMA the id check overlaps with)."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

import datagen
from conftest import ROOT

pytestmark = pytest.mark.gpu
CONVERT = os.path.join(ROOT, "oracle", "_ref", "convert")
TRANSPOSE = os.path.join(ROOT, "oracle", "_ref", "transpose")


def read_binary_raw(capi, prefix):
    rows, err = capi.HostRows(), C.create_string_buffer(512)
    rc = capi.load().fmx_read_binary(os.fsencode(prefix), C.byref(rows), err, len(err))
    assert rc == capi.FMX_OK, err.value
    return rows


def upload_raw(capi, h, slot, rows):
    rc = h.lib.fmx_upload_rows(h.h, slot, rows.entries, rows.row_ptr, rows.target, rows.n_rows, rows.nnz)
    assert rc == capi.FMX_OK, h.lib.fmx_last_error(h.h)


@pytest.mark.skipif(not (os.path.exists(CONVERT) and os.path.exists(TRANSPOSE)), reason="oracle/_ref/convert / transpose not built (needs /root/reference)")
@pytest.mark.parametrize("which", ["x", "xt"])
def test_reference_binary_files_to_learners(tmp_path, oracle, which):
    from libfm_amd import capi
    O = oracle
    n, nnz, n_rows, k = 1600, 8, 3000, 16
    ent, rp, y = datagen.onehot_fields(n, nnz, n_rows, seed=21, classification=False)
    txt, pre = str(tmp_path / "d.libfm"), str(tmp_path / "d")
    O.Data(ent, rp, y).write_libsvm(txt)
    subprocess.run([CONVERT, "--ifile", txt, "--ofilex", pre + ".x", "--ofiley", pre + ".y"], check=True, capture_output=True)
    subprocess.run([TRANSPOSE, "--ifile", pre + ".x", "--ofile", pre + ".xt"], check=True, capture_output=True)
    if which == "xt":                               # what an als / mcmc run keeps on disk (libfm.cpp:143-147): the transpose alone
        os.remove(pre + ".x")
    rows = read_binary_raw(capi, pre)
    assert rows.flags & 1, "fmx_read_binary must hand out page-locked buffers on a GPU box (FMX_HOST_PINNED)"
    assert rows.n_rows == n_rows and rows.nnz == len(ent)
    d = O.Data(ent, rp, y)
    lo, hi = float(y.min()), float(y.max())
    # ---- one SGD epoch (batch rule) on the rows as they came from the file
    m = O.Model(n, k, True, True, 0.0, 0.0, 0.002)
    m.v[:] = O.init_values(3, n, k, 0.05)
    h = capi.Handle(n, k, True, True, capi.TASK_REGRESSION, 0.0, 0.0, 0.002, 0.005, lo, hi, device=0)
    h.set_params(m.w0, m.w, m.v)
    upload_raw(capi, h, 0, rows)
    e2, r2, y2 = h.download_rows(0)
    assert np.array_equal(e2, ent) and np.array_equal(r2, rp) and np.array_equal(y2, y)     # (one-hot rows: ids already ascending)
    h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 512, 32, 0, 2)
    O.sgd_epoch_minibatch(m, d, 0, 0.005, lo, hi, 512, 32, bias_lag=2)
    w0, w, v = h.get_params()
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=2e-5)
    h.close()
    # ---- one ALS sweep (device transpose of the uploaded rows)
    m = O.Model(n, k, True, True, 0.1, 1.0, 5.0)
    m.v[:] = O.init_values(4, n, k, 0.1)
    m.w[:] = O.init_values(5, n, 1, 0.1)[0]
    h = capi.Handle(n, k, True, True, capi.TASK_REGRESSION, 0.1, 1.0, 5.0, 0.0, lo, hi, device=0)
    h.set_params(m.w0, m.w, m.v)
    upload_raw(capi, h, 0, rows)
    h.als_begin(0)
    h.als_sweep(1.0, 5.0)
    h.als_end()
    O.als_learn(m, d, d, 0, 1, 1.0, 5.0, lo, hi)
    w0, w, v = h.get_params()
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=2e-5)
    assert abs(w0 - m.w0) <= 1e-4 * abs(m.w0) + 2e-5
    h.close()
    capi.load().fmx_free_host_rows(C.byref(rows))


def test_upload_from_page_locked_buffers_is_a_dma(tmp_path):
    """64 MB of entries written in the reference's format, read back by fmx_read_binary (page-locked) and uploaded, against the
    same rows uploaded from pageable numpy memory: the pinned path must not lose (it runs at PCIe speed under the id check)."""
    from libfm_amd import capi, data as D
    n, nnz, n_rows = 1 << 20, 16, 1 << 19
    ent, rp, y = datagen.onehot_fields(n, nnz, n_rows, seed=2)
    pre = str(tmp_path / "big")
    D.write_binary(pre, ent, rp, y, num_cols=n)
    rows = read_binary_raw(capi, pre)
    assert rows.flags & 1
    h = capi.Handle(n, 8, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.0, 0.01, -1.0, 1.0, device=0)

    def best(fn):
        ts = []
        for _ in range(4):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return min(ts)
    t_pinned = best(lambda: upload_raw(capi, h, 0, rows))
    t_pageable = best(lambda: h.upload_rows(1, ent, rp, y))
    e2, r2, y2 = h.download_rows(0)
    assert np.array_equal(e2, ent) and np.array_equal(r2, rp)
    h.close()
    capi.load().fmx_free_host_rows(C.byref(rows))
    print("upload of %d MB: page-locked %.1f ms, pageable %.1f ms" % (ent.nbytes >> 20, t_pinned * 1e3, t_pageable * 1e3))
    assert t_pinned <= 1.5 * t_pageable
