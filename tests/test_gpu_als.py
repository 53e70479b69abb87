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


def _split(monkeypatch, v):
    """fmx_config::als_split_min for the handles created from here on: "0" = never split (fused draws), "1" = every level, n = levels of >= n entries"""
    from libfm_amd import capi as _c
    monkeypatch.setattr(_c, "ALS_SPLIT_MIN", _c.ALS_SPLIT_NEVER if str(v) == "0" else int(v))

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


@pytest.fixture(params=["fused", "split"])
def draw_form(request, monkeypatch):
    """fused: column sums, draw and {e, q} update in one launch per (family, level); split: the update as a row-ordered
    stream (k_als_rows) -- the library's choice for levels of >= fmx_config::als_split_min entries, forced
    here for every level so that the small fixtures run it too (duplicate ids in a row, ragged rows, groups, probit)."""
    _split(monkeypatch, "1" if request.param == "split" else "0")
    return request.param


@pytest.mark.parametrize("name", CASES)
def test_als_matches_reference(oracle, name, draw_form):
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


def test_als_field_data_bigger_than_fixture(oracle, draw_form):
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


def test_als_mid_size_against_oracle(oracle, draw_form):
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


@pytest.mark.parametrize("task,do_sample", [(0, False), (1, True)])
def test_split_step_is_the_fused_step(oracle, monkeypatch, task, do_sample):
    """the row-ordered update performs the same fp64 operations per row as the fused kernel's second pass (the {old, new} pair
    travels as two fp32 numbers, both exact), so the two forms agree to fp64 rounding (the compiler contracts the
    multiply-adds of the two kernels differently) -- fp32-identical parameters, also for a sampled chain."""
    from libfm_amd import learner as L
    n, nnz, k = 60000, 12, 8
    ent, rp, y = datagen.onehot_fields(n, nnz, 30000, seed=31, classification=bool(task))
    ent2, rp2, y2 = datagen.onehot_fields(n, nnz, 2000, seed=32, classification=bool(task))
    res = []
    for split_min in ("0", "1", "20000"):                      # never / always / the big levels only (30 000 entries each)
        _split(monkeypatch, split_min)
        fm = L.FMModel()
        fm.num_attribute, fm.num_factor = n, k
        fm.w0, fm.w, fm.v = 0.1, oracle.init_values(5, n, 1, 0.1)[0].copy(), oracle.init_values(4, n, k, 0.1).copy()
        l = L.FMLearnALS()
        l.fm, l.task, l.num_iter, l.min_target, l.max_target = fm, task, 3, float(y.min()), float(y.max())
        l.w_lambda, l.v_lambda, l.do_sample, l.seed = 1.0, 4.0, do_sample, 99
        l.out = io.StringIO()
        l.init()
        l.learn(L.Data(ent, rp, y), L.Data(ent2, rp2, y2))
        res.append((l.fm.w0, l.fm.w.copy(), l.fm.v.copy(), l.predict(L.Data(ent2, rp2, y2)).copy()))
        l.close()
    assert np.abs(res[0][2] - oracle.init_values(4, n, k, 0.1)).max() > 1e-3
    for other in res[1:]:
        assert abs(other[0] - res[0][0]) < 1e-10
        np.testing.assert_allclose(other[1], res[0][1], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(other[2], res[0][2], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(other[3], res[0][3], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("k", [100, 200, 300, 700])
def test_als_wide_rows_against_oracle(oracle, k, draw_form):
    """KP = 128 / 256 (two / four floats per lane; the re-prediction hands 8 / 4 rows of q_f per wavefront through LDS)"""
    from libfm_amd import learner as L
    n, nnz = 1200, 6
    ent, rp, y = datagen.onehot_fields(n, nnz, 700, seed=41 + k, classification=False)
    ent2, rp2, y2 = datagen.onehot_fields(n, nnz, 100, seed=42 + k, classification=False)
    m = oracle.Model(n, k, True, True, 0.0, 1.0, 6.0)
    m.v[:] = oracle.init_values(14, n, k, 0.05)
    m.w[:] = oracle.init_values(15, n, 1, 0.05)[0]
    lo, hi = float(y.min()), float(y.max())
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor = n, k
    fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
    l = L.FMLearnALS()
    l.fm, l.task, l.num_iter, l.min_target, l.max_target, l.w_lambda, l.v_lambda = fm, 0, 2, lo, hi, 1.0, 6.0
    l.out = io.StringIO()
    l.init()
    l.learn(L.Data(ent, rp, y), L.Data(ent2, rp2, y2))
    pred, metric = oracle.als_learn(m, oracle.Data(ent, rp, y), oracle.Data(ent2, rp2, y2), 0, 2, 1.0, 6.0, lo, hi)
    np.testing.assert_allclose(l.fm.v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.fm.w, m.w, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.pred_this, pred, rtol=1e-4, atol=5e-5)
    l.close()
