"""GPU: the ALS learner (fm_learn_mcmc with do_sample = 0) against the REAL reference's results (golden fixtures
produced by oracle/_ref/ref_harness als) and the pinned oracle.  Tolerance 1e-4 relative (fp32 parameter storage;
e/q caches and all reductions are fp64 on the device)."""
import io

import numpy as np
import pytest

import datagen
from common import Golden
from conftest import golden_cases

pytestmark = pytest.mark.gpu
CASES = [c for c in golden_cases() if c.startswith("als_")]


def make_learner(g, oracle):
    from libfm_amd import learner as L
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = g.n, g.k, bool(g.k0), bool(g.k1)
    fm.reg0, fm.regw, fm.regv = g.reg
    m = g.model(oracle, "init")
    fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
    l = L.FMLearnALS()
    l.fm, l.task, l.num_iter = fm, g.task, g.iters
    l.min_target, l.max_target = g.min_target, g.max_target
    l.w_lambda, l.v_lambda = g.reg[1], g.reg[2]
    if "group" in g.z.files:                                   # `-meta` + per-group lambdas (libfm.cpp:353-363)
        l.groups, l.w_lambda, l.v_lambda = g.z["group"], g.z["w_lambda_g"], g.z["v_lambda_g"]
    l.out = io.StringIO()
    return L, l


@pytest.mark.parametrize("name", CASES)
def test_als_matches_reference(oracle, name):
    g = Golden(name)
    L, l = make_learner(g, oracle)
    z = g.z
    train = L.Data(z["train_entries"], z["train_row_ptr"], g.train_target)
    test = L.Data(z["test_entries"], z["test_row_ptr"], g.test_target)
    l.init()
    l.learn(train, test)
    assert abs(l.fm.w0 - float(z["final_w0"])) <= 1e-4 * abs(float(z["final_w0"])) + 2e-5
    np.testing.assert_allclose(l.fm.w, z["final_w"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.fm.v, z["final_v"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.predict(test), z["pred_out"], rtol=1e-4, atol=5e-5)
    assert "#Iter=" in l.out.getvalue()
    l.close()


def test_als_field_data_bigger_than_fixture(oracle):
    """one-hot field rows (level = field): 16 fields, k = 32, compared with the pinned oracle."""
    from libfm_amd import learner as L
    n, nnz = 3200, 16
    ent, rp, y = datagen.onehot_fields(n, nnz, 2000, seed=9, classification=False)
    ent2, rp2, y2 = datagen.onehot_fields(n, nnz, 500, seed=10, classification=False)
    k = 32
    m = oracle.Model(n, k, True, True, 0.1, 1.0, 5.0)
    m.v[:] = oracle.init_values(4, n, k, 0.1)
    m.w[:] = oracle.init_values(5, n, 1, 0.1)[0]
    lo, hi = float(y.min()), float(y.max())
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.reg0, fm.regw, fm.regv = n, k, 0.1, 1.0, 5.0
    fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
    l = L.FMLearnALS()
    l.fm, l.task, l.num_iter, l.min_target, l.max_target, l.w_lambda, l.v_lambda = fm, 0, 3, lo, hi, 1.0, 5.0
    l.out = io.StringIO()
    l.init()
    l.learn(L.Data(ent, rp, y), L.Data(ent2, rp2, y2))
    pred, metric = oracle.als_learn(m, oracle.Data(ent, rp, y), oracle.Data(ent2, rp2, y2), 0, 3, 1.0, 5.0, lo, hi)
    np.testing.assert_allclose(l.fm.v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.fm.w, m.w, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.pred_this, pred, rtol=1e-4, atol=5e-5)
    assert l.log[-1]["levels"] == nnz
    np.testing.assert_allclose([x["train"] for x in l.log], metric, rtol=1e-4)
    l.close()


def test_als_mid_size_against_oracle(oracle):
    """50 000 rows x 16 one-hot fields over 200 000 features, k = 16, classification (probit targets): thousands of
    columns per level and G = 4 lanes per column -- a different launch regime from the small fixtures."""
    from libfm_amd import learner as L
    n, nnz, k = 200000, 16, 16
    ent, rp, y = datagen.onehot_fields(n, nnz, 50000, seed=21)
    ent2, rp2, y2 = datagen.onehot_fields(n, nnz, 5000, seed=22)
    m = oracle.Model(n, k, True, True, 0.2, 2.0, 8.0)
    m.v[:] = oracle.init_values(6, n, k, 0.1)
    m.w[:] = oracle.init_values(7, n, 1, 0.1)[0]
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.reg0, fm.regw, fm.regv = n, k, 0.2, 2.0, 8.0
    fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
    l = L.FMLearnALS()
    l.fm, l.task, l.num_iter, l.min_target, l.max_target, l.w_lambda, l.v_lambda = fm, 1, 2, -1.0, 1.0, 2.0, 8.0
    l.out = io.StringIO()
    l.init()
    l.learn(L.Data(ent, rp, y), L.Data(ent2, rp2, y2))
    pred, metric = oracle.als_learn(m, oracle.Data(ent, rp, y), oracle.Data(ent2, rp2, y2), 1, 2, 2.0, 8.0, -1.0, 1.0)
    np.testing.assert_allclose(l.fm.v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.fm.w, m.w, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.pred_this, pred, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose([x["train"] for x in l.log], metric, rtol=1e-4)
    l.close()
