"""GPU: `-method sgda` (fm_learn_sgd_element_adapt_reg) against the REAL reference's results (fixtures produced by
oracle/_ref/ref_harness sgda): final parameters, the LEARNED regularisation values and the -out predictions."""
import io

import numpy as np
import pytest

from common import Golden
from conftest import golden_cases

pytestmark = pytest.mark.gpu
CASES = [c for c in golden_cases() if c.startswith("sgda_")]


@pytest.mark.parametrize("name", CASES)
def test_sgda_matches_reference(oracle, name):
    from libfm_amd import learner as L
    g = Golden(name)
    z = g.z
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = g.n, g.k, bool(g.k0), bool(g.k1)
    m = g.model(oracle, "init")
    fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
    vt = z["val_target"].copy()
    if g.task == 1:
        vt = np.where(vt <= 0, -1.0, 1.0).astype(np.float32)
    l = L.FMLearnSGDA()
    l.fm, l.task, l.num_iter, l.learn_rate = fm, g.task, g.iters, g.lr
    l.min_target, l.max_target = g.min_target, g.max_target
    l.validation = L.Data(z["val_entries"], z["val_row_ptr"], vt)
    grouped = "group" in z.files                                # `-meta` attribute groups: reg_w(g), reg_v(g,f)
    if grouped:
        l.groups = z["group"]
    l.out = io.StringIO()
    train = L.Data(z["train_entries"], z["train_row_ptr"], g.train_target)
    test = L.Data(z["test_entries"], z["test_row_ptr"], g.test_target)
    l.init()
    l.learn(train, test)
    np.testing.assert_allclose(fm.v, z["final_v"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(fm.w, z["final_w"], rtol=1e-4, atol=2e-5)
    assert abs(fm.w0 - float(z["final_w0"])) <= 1e-4 * abs(float(z["final_w0"])) + 2e-5
    if grouped:
        regs = z["regs"].reshape(-1, 1 + g.k)                   # [G][1 + k]
        assert regs.shape[0] == int(z["group"].max()) + 1 > 1
        np.testing.assert_allclose(l.reg_w, regs[:, 0], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(l.reg_v, regs[:, 1:], rtol=1e-3, atol=1e-7)
        assert {"regw[1]", "regv[1,0]"} <= set(l.log[-1])
    else:
        np.testing.assert_allclose(l.reg_w, z["regs"][0], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(l.reg_v, z["regs"][1:], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(l.predict(test), z["pred_out"], rtol=1e-4, atol=5e-5)
    l.close()


# ---------------------------------------------------------------------------------------------
# the batch form (fmx_sgda_epoch_minibatch) against its oracle restatement (fmo_sgda_epoch_minibatch, itself pinned to the
# reference at batch 1 by tests/test_oracle_sgda_golden.py): parameters 1e-4, learned regularisation 1e-3
# ---------------------------------------------------------------------------------------------
def _val(g):
    vt = g.z["val_target"].copy()
    return np.where(vt <= 0, -1.0, 1.0).astype(np.float32) if g.task == 1 else vt


@pytest.mark.parametrize("name,batch,chunk", [("sgda_reg_ml", 1, 1), ("sgda_reg_ml", 16, 4), ("sgda_reg_ml", 100, 8),
                                              ("sgda_cls_fields", 32, 8), ("sgda_cls_fields", 7, 1),
                                              ("sgda_reg_ml_groups", 16, 4), ("sgda_cls_fields_groups", 50, 10)])
def test_sgda_batch_form_matches_its_rule(oracle, name, batch, chunk):
    from libfm_amd import capi
    O = oracle
    g = Golden(name)
    z = g.z
    group = z["group"] if "group" in z.files else None
    m = g.model(O, "init")
    m.reg0 = m.regw = m.regv = 0.0
    tr = g.data(O, "train")
    va = O.Data(z["val_entries"], z["val_row_ptr"], _val(g))
    h = capi.Handle(g.n, g.k, g.k0, g.k1, g.task, 0.0, 0.0, 0.0, g.lr, g.min_target, g.max_target)
    h.set_params(m.w0, m.w, m.v)
    if group is not None:
        h.set_groups(group)
    h.upload_rows(0, tr.entries, tr.row_ptr, tr.target)
    h.upload_rows(1, va.entries, va.row_ptr, va.target)
    h.sgda_begin()
    for i in range(g.iters):
        h.sgda_epoch_minibatch(0, 1, i > 0, batch, chunk)
    reg = h.sgda_get_reg()
    w0, w, v = h.get_params()
    h.sgda_end()
    h.close()
    st = O.sgda_learn(m, tr, va, g.task, g.lr, g.min_target, g.max_target, g.iters, group, batch=batch, w0_chunk=chunk)
    assert abs(w0 - m.w0) <= 1e-4 * abs(m.w0) + 2e-5
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(reg[:, 0], st.reg_w, rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(reg[:, 1:], st.reg_v[:, :g.k], rtol=1e-3, atol=1e-7)
    if batch == 1 and not g.has_duplicate_ids():               # ... and at batch 1 that rule is the reference itself
        np.testing.assert_allclose(v, z["final_v"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(w, z["final_w"], rtol=1e-4, atol=2e-5)


def test_sgda_batch_form_mid_size(oracle):
    """20 000 train / 5 000 validation rows over 6 400 one-hot features, k = 16, batch 1 024: thousands of features collide in
    every batch, the validation pointer wraps around four times per epoch"""
    from libfm_amd import capi
    import datagen
    O = oracle
    n, nnz, k, lr = 6400, 8, 16, 0.002
    ent, rp, y = datagen.onehot_fields(n, nnz, 20000, seed=3, classification=False)
    ev, rv, yv = datagen.onehot_fields(n, nnz, 5000, seed=4, classification=False)
    lo, hi = float(y.min()), float(y.max())
    m = O.Model(n, k, True, True, 0.0, 0.0, 0.0)
    m.v[:] = O.init_values(9, n, k, 0.05)
    tr, va = O.Data(ent, rp, y), O.Data(ev, rv, yv)
    h = capi.Handle(n, k, True, True, 0, 0.0, 0.0, 0.0, lr, lo, hi)
    h.set_params(m.w0, m.w, m.v)
    h.upload_rows(0, ent, rp, y)
    h.upload_rows(1, ev, rv, yv)
    h.sgda_begin()
    for i in range(3):
        h.sgda_epoch_minibatch(0, 1, i > 0, 1024, 64)
    reg = h.sgda_get_reg()
    w0, w, v = h.get_params()
    h.sgda_end()
    h.close()
    st = O.sgda_learn(m, tr, va, 0, lr, lo, hi, 3, None, batch=1024, w0_chunk=64)
    assert abs(w0 - m.w0) <= 1e-4 * abs(m.w0) + 2e-5
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(reg[0, 0], st.reg_w[0], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(reg[0, 1:], st.reg_v[0, :k], rtol=2e-3, atol=1e-7)
