"""CPU: pins the C restatement (oracle/fm_oracle.c) against the REAL reference's outputs.

The fixtures in tests/golden/ were produced by oracle/_ref/ref_harness (the reference's own fm_model,
fm_SGD, fm_learn_sgd_element classes compiled from /root/reference) -- see tests/golden/make_golden.py.
Bar: bit-exact fp64 (the restatement follows the reference's loop nests and summation order)."""
import numpy as np
import pytest

from common import Golden
from conftest import golden_cases

CASES = [c for c in golden_cases() if c.startswith("sgd_")]


@pytest.mark.parametrize("name", CASES)
def test_online_sgd_bit_exact(oracle, name):
    O = oracle
    g = Golden(name)
    m = g.model(O, "init")
    tr, te = g.data(O, "train"), g.data(O, "test")
    evals = []
    for _ in range(g.iters):
        O.sgd_epoch_online(m, tr, g.task, g.lr, g.min_target, g.max_target)
        evals.append([O.evaluate(m, tr, g.task, g.min_target, g.max_target)[0],
                      O.evaluate(m, te, g.task, g.min_target, g.max_target)[0]])
    assert m.w0 == float(g.z["final_w0"])
    assert np.array_equal(m.w, g.z["final_w"])
    assert np.array_equal(m.v, g.z["final_v"])
    assert np.array_equal(np.array(evals), g.z["eval"])
    assert np.array_equal(O.predict_raw(m, te), g.z["pred_raw"])
    assert np.array_equal(O.predict_out(m, te, g.task, g.min_target, g.max_target), g.z["pred_out"])


@pytest.mark.parametrize("name", CASES)
def test_predict_on_reference_final_params(oracle, name):
    """forward parity on fixed parameters: fm_model::predict (fm_model.h:105-127)."""
    O = oracle
    g = Golden(name)
    m = g.model(O, "final")
    assert np.array_equal(O.predict_raw(m, g.data(O, "test")), g.z["pred_raw"])


@pytest.mark.parametrize("name", CASES)
def test_minibatch_rule_collapses_to_reference_at_batch_1(oracle, name):
    O = oracle
    g = Golden(name)
    if g.has_duplicate_ids():
        pytest.skip("rows with a repeated id: the batch rule uses batch-start v for both occurrences")
    m = g.model(O, "init")
    tr = g.data(O, "train")
    for _ in range(g.iters):
        O.sgd_epoch_minibatch(m, tr, g.task, g.lr, g.min_target, g.max_target, 1, 1)
    # not bit-exact: the batch rule forms p = w0 + rest_e (w0 added last) while fm_model.h:107-115 starts
    # the sum with w0, so p can differ in the last ulp; everything else is the same arithmetic.
    np.testing.assert_allclose(m.w0, float(g.z["final_w0"]), rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(m.w, g.z["final_w"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(m.v, g.z["final_v"], rtol=1e-11, atol=1e-14)


def test_minibatch_rule_is_close_to_online_for_small_batches(oracle):
    """sanity of the restated rule: B=8 stays near the online trajectory on a smooth problem."""
    O = oracle
    g = Golden("sgd_reg_ml")
    m1, m2 = g.model(O, "init"), g.model(O, "init")
    tr = g.data(O, "train")
    for _ in range(g.iters):
        O.sgd_epoch_online(m1, tr, g.task, g.lr, g.min_target, g.max_target)
        O.sgd_epoch_minibatch(m2, tr, g.task, g.lr, g.min_target, g.max_target, 8, 1)
    r1 = O.evaluate(m1, tr, g.task, g.min_target, g.max_target)[0]
    r2 = O.evaluate(m2, tr, g.task, g.min_target, g.max_target)[0]
    assert abs(r1 - r2) < 0.02 * r1


def test_synth_generator_definition(oracle):
    O = oracle
    d = O.synth_rows(123, 0, 100, 8, 800)
    ids = d.entries["id"].reshape(100, 8)
    fs = 100
    assert ((ids // fs) == np.arange(8)[None, :]).all()          # field t owns [t*fs,(t+1)*fs)
    assert set(np.unique(d.target)) <= {-1.0, 1.0}
    d2 = O.synth_rows(123, 50, 50, 8, 800)                        # row0 offset is consistent
    assert np.array_equal(d2.entries, d.entries[50 * 8:])
    assert O.lib().fmo_init_value(7, 3, 2, 0.5) == O.init_values(7, 8, 4, 0.5)[2, 3]


def _numpy_minibatch_rule(m, d, task, lr, lo, hi, batch, chunk, lag):
    """independent numpy restatement of the batch rule with a bias lag of `lag` batches (fm_oracle.h): used only to pin
    the C oracle's bookkeeping of WHICH bias a batch's multipliers see."""
    ids, x = d.entries["id"].astype(np.int64), d.entries["value"].astype(np.float64)
    rp = d.row_ptr.astype(np.int64)

    def mult_of(p, y):
        if task == 0:
            return -(y - np.clip(p, lo, hi))
        return -y * (1.0 - 1.0 / (1.0 + np.exp(-y * p)))
    w0_start = []
    for r0 in range(0, d.n_rows, batch):
        nb = min(batch, d.n_rows - r0)
        w0_start.append(m.w0)
        S, rest = np.zeros((nb, m.k)), np.zeros(nb)
        for e in range(nb):
            sl = slice(rp[r0 + e], rp[r0 + e + 1])
            vx = m.v[:, ids[sl]] * x[sl]
            S[e] = vx.sum(axis=1)
            rest[e] = (m.w[ids[sl]] * x[sl]).sum() * m.k1 + 0.5 * (S[e] ** 2 - (vx ** 2).sum(axis=1)).sum()
        y = d.target[r0:r0 + nb].astype(np.float64)
        b = len(w0_start) - 1
        w0_used = w0_start[max(b - lag + 1, 0)] if lag else None
        mult = np.zeros(nb)
        for c0 in range(0, nb, chunk):
            sl = slice(c0, min(c0 + chunk, nb))
            me = mult_of(m.w0 + rest[sl], y[sl])
            mult[sl] = me if not lag else mult_of(w0_used + rest[sl], y[sl])
            m.w0 -= lr * (me + m.reg0 * m.w0).sum()
        dw, dv = np.zeros_like(m.w), np.zeros_like(m.v)
        for e in range(nb):
            for i in range(rp[r0 + e], rp[r0 + e + 1]):
                j, xv = ids[i], x[i]
                dw[j] += -lr * (mult[e] * xv + m.regw * m.w[j])
                dv[:, j] += -lr * (mult[e] * (S[e] * xv - m.v[:, j] * xv * xv) + m.regv * m.v[:, j])
        m.w += dw * m.k1
        m.v += dv


@pytest.mark.parametrize("lag", [0, 1, 2, 3])
@pytest.mark.parametrize("name", ["sgd_cls_zipf_k32", "sgd_reg_ml"])
def test_bias_lag_depth_bookkeeping(oracle, name, lag):
    O = oracle
    g = Golden(name)
    tr = g.data(O, "train")
    rows = min(tr.n_rows, 400)
    sub = O.Data(tr.entries[: int(tr.row_ptr[rows])], tr.row_ptr[: rows + 1], tr.target[:rows])
    a, b = g.model(O, "init"), g.model(O, "init")
    for _ in range(2):
        O.sgd_epoch_minibatch(a, sub, g.task, g.lr, g.min_target, g.max_target, 50, 10, bias_lag=lag)
        _numpy_minibatch_rule(b, sub, g.task, g.lr, g.min_target, g.max_target, 50, 10, lag)
    np.testing.assert_allclose(a.w0, b.w0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a.w, b.w, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a.v, b.v, rtol=1e-9, atol=1e-12)
    if lag >= 2:                                                  # and the depth does change the result
        c = g.model(O, "init")
        for _ in range(2):
            O.sgd_epoch_minibatch(c, sub, g.task, g.lr, g.min_target, g.max_target, 50, 10, bias_lag=1)
        assert np.abs(c.v - a.v).max() > 0
