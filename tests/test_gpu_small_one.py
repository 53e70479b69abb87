"""GPU: small batches of the one-pass minibatch rule as ONE launch per batch across all dies (libfm_amd/csrc/fmx_small_kernels.h k_small_one;
the default for batches of up to 1 024 rows; FMX_SMALL_ONE=0 at fmx_create: two launches) -- examples, deferred (frequent) features and the bias recurrence of a batch exchange through tagged 8-byte
slots instead of a launch boundary (round-5 verdict item 2; fm_learn_sgd_element.h:56-67 is that chain at batch 1).

It is the same batch rule as the two launches per batch (k_fused<EXACT> + k_apply_seg_scan): held against the oracle at 1e-4 and against
the two-launch path at fp32 rounding, for both tasks, bias lags 1 and 2, ragged rows, k = 64 and k = 100 (128-float rows), and k = 8 / 16 / 24 / 32 (several rows per wave-wide load)."""
import numpy as np
import pytest

import datagen as DG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import capi as c
    if c.load().fmx_device_count() == 0:
        pytest.fail("no HIP device: the GPU tests must run on the MI355X box")
    return c


def ragged(ent, rp, y, seed, keep=0.8, values=False):
    """drop entries at random (rows of different lengths: the row_ptr form of the entry list) and, optionally, give them real values"""
    rng = np.random.default_rng(seed)
    n_rows = len(y)
    keep_mask = rng.random(len(ent)) < keep
    keep_mask[rp[:-1].astype(np.int64)] = True                    # (no empty rows here; the empty-row case has its own test)
    rows = np.repeat(np.arange(n_rows), np.diff(rp).astype(np.int64))
    e2 = ent[keep_mask].copy()
    if values:
        e2["value"] = np.round(rng.uniform(0.25, 1.5, len(e2)), 3).astype(np.float32)
    cnt = np.bincount(rows[keep_mask], minlength=n_rows)
    rp2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    return e2, rp2, y


def run_case(capi, oracle, monkeypatch, k, task, lag, rows, make_ragged, epochs=2, lr=0.01, seed=5):
    O = oracle
    e, rp, y, n = DG.criteo_shaped(rows, seed, cat_ids=2000, classification=(task == 1))
    if make_ragged:
        e, rp, y = ragged(e, rp, y, seed + 1, values=True)
    d = O.Data(e, rp, y)
    m = O.Model(n, k, True, True, 0.0, 0.0005, 0.001)
    m.v[:] = O.init_values(1, n, k, 0.05)
    m.w0 = 0.02
    out = {}
    for tag in ("one", "two_launches"):
        monkeypatch.setenv("FMX_SMALL_ONE", "1" if tag == "one" else "0")   # (read by fmx_create; the default is one launch per batch)
        h = capi.Handle(n, k, True, True, task, 0.0, 0.0005, 0.001, lr, -3.0, 3.0, device=0)
        h.set_params(m.w0, m.w, m.v)
        h.upload_rows(0, e, rp, y)
        B = h.sgd_batch_info(0).batch
        assert 64 <= B <= 4096 and rows // B >= 4
        for _ in range(epochs):
            st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, lag)
            assert st.batch_used == B and st.batches == (rows + B - 1) // B
            assert bool(st.status & capi.STAT_SMALL_ONE) == (tag == "one"), (tag, st.status)
            assert st.deferred_features > rows
        out[tag] = (h.get_params(), h.predict(0, rows), st.w0_chunk_used, B)
        h.close()
    (w0, w, v), pred, chunk, B = out["one"]
    for _ in range(epochs):
        O.sgd_epoch_minibatch(m, d, task, lr, -3.0, 3.0, B, chunk, bias_lag=lag)
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=2e-5)
    assert abs(w0 - m.w0) <= 1e-4 * abs(m.w0) + 2e-5
    np.testing.assert_allclose(pred, O.predict_raw(m, d), rtol=1e-4, atol=5e-5)
    (w0b, wb, vb), predb, _, _ = out["two_launches"]
    np.testing.assert_allclose(v, vb, rtol=2e-5, atol=2e-7)      # the same rule; one-hot rows take the fused form of fm_sgd.h:47-49 (a few ulp)
    np.testing.assert_allclose(w, wb, rtol=2e-5, atol=2e-7)
    assert abs(w0 - w0b) <= 2e-5 * abs(w0b) + 2e-7


@pytest.mark.parametrize("k,task,lag,make_ragged", [(64, 1, 2, False), (64, 1, 1, True), (64, 0, 2, True), (100, 1, 2, False), (100, 0, 1, True),
                                                     (8, 1, 2, False), (8, 0, 1, True), (16, 1, 2, True), (24, 1, 2, False), (32, 0, 2, True)])
def test_one_launch_batches_is_the_batch_rule(capi, oracle, monkeypatch, k, task, lag, make_ragged):
    run_case(capi, oracle, monkeypatch, k, task, lag, 6000, make_ragged, lr=0.01 if task == 1 else 0.002)


def test_one_launch_batches_many_batches_and_a_short_last_one(capi, oracle, monkeypatch):
    """20 011 rows: ~60 batches and a last batch of a few rows (fewer examples than wavefronts; segments of one batch only)"""
    run_case(capi, oracle, monkeypatch, 64, 1, 2, 20011, False, epochs=1)


def test_one_launch_batches_without_bias_and_without_linear_terms(capi, oracle, monkeypatch):
    """k0 = 0 (no recurrence: every wavefront takes items) and k1 = 0 (no w gathers)"""
    O = oracle
    monkeypatch.setenv("FMX_SMALL_ONE", "1")
    rows, k = 5000, 64
    e, rp, y, n = DG.criteo_shaped(rows, 11, cat_ids=2000)
    d = O.Data(e, rp, y)
    for k0, k1 in ((False, True), (True, False)):
        m = O.Model(n, k, k0, k1, 0.0, 0.0005, 0.001)
        m.v[:] = O.init_values(2, n, k, 0.05)
        h = capi.Handle(n, k, k0, k1, 1, 0.0, 0.0005, 0.001, 0.01, -1.0, 1.0, device=0)
        h.set_params(m.w0, m.w, m.v)
        h.upload_rows(0, e, rp, y)
        st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, 2)
        assert st.status & capi.STAT_SMALL_ONE
        O.sgd_epoch_minibatch(m, d, 1, 0.01, -1.0, 1.0, st.batch_used, st.w0_chunk_used, bias_lag=2)
        w0, w, v = h.get_params()
        np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=2e-5)
        assert abs(w0 - m.w0) <= 1e-4 * abs(m.w0) + 2e-5
        h.close()
