"""CPU: what decides fmx_sgd_opts::batch = 0 -- the batch rule (oracle fmo_sgd_epoch_minibatch_ex) follows the reference's online
loop (fm_learn_sgd_element.h:56-67 / fm_sgd.h:33-51, restated and pinned bit-exact in test_oracle_golden.py) while
    gain = learn_rate * curvature * batch * C <= 1,     C = collision mass of the rows (tests/datagen.py collision_mass)
and diverges beyond a gain of ~2: two examples of a batch that share features push them in the same direction from the same
frozen state.  Checked on the Criteo-shaped rows of BASELINE configs[2] (C ~ 1), a milder Zipf set and the ML-100K shape."""
import numpy as np
import pytest

import datagen as DG


def logloss(O, m, d):
    p = O.predict_raw(m, d)
    with np.errstate(over="ignore", invalid="ignore"):
        return float(np.mean(np.log1p(np.exp(-d.target.astype(np.float64) * p))))


def sets(O):
    e, rp, y, n = DG.criteo_shaped(36000, 5, cat_ids=2000)
    z = 39
    yield ("criteo_shaped", O.Data(e[:30000 * z], rp[:30001], y[:30000]), O.Data(e[30000 * z:], rp[30000:] - rp[30000], y[30000:]),
           n, 1, 0.01, -1.0, 1.0)
    e, rp, y = DG.onehot_fields(16 * 1500, 16, 36000, 3, zipf=0.8)
    yield ("zipf_0.8", O.Data(e[:30000 * 16], rp[:30001], y[:30000]), O.Data(e[30000 * 16:], rp[30000:] - rp[30000], y[30000:]),
           16 * 1500, 1, 0.02, -1.0, 1.0)
    e, rp, y = DG.movielens_shaped(943, 1682, 50000, 42)
    yield ("ml100k_shaped", O.Data(e[:40000 * 2], rp[:40001], y[:40000]), O.Data(e[40000 * 2:], rp[40000:] - rp[40000], y[40000:]),
           2625, 0, 0.01, 1.0, 5.0)


def test_batch_rule_follows_the_online_loop_up_to_gain_one_and_diverges_beyond(oracle):
    O = oracle
    for name, tr, te, n, task, lr, lo, hi in sets(O):
        C = DG.collision_mass(tr.entries, tr.n_rows, n)
        chunk = 256 if task == 1 else 64
        B_ok = DG.stable_batch(lr, task, C, default=1 << 30)
        assert lr * (1.0 if task == 0 else 0.25) * B_ok * C <= 1.0 < lr * (1.0 if task == 0 else 0.25) * 2 * B_ok * C

        def run(batch):
            m = O.Model(n, 8, True, True, 0.0, 0.0, 0.001)
            m.v[:] = O.init_values(1, n, 8, 0.01)
            for _ in range(3):
                if batch is None:
                    O.sgd_epoch_online(m, tr, task, lr, lo, hi)
                else:
                    O.sgd_epoch_minibatch(m, tr, task, lr, lo, hi, batch, min(chunk, batch), bias_lag=2)
            return logloss(O, m, te) if task == 1 else O.evaluate(m, te, task, lo, hi)[0]
        ref = run(None)
        ok = run(min(B_ok, tr.n_rows))
        assert abs(ok - ref) <= 0.02, (name, B_ok, ok, ref)              # the library's batch: the reference's result
        if 8 * B_ok <= tr.n_rows:
            bad = run(8 * B_ok)                                           # gain in (4, 8]: the rule has left the trajectory
            assert not np.isfinite(bad) or bad > ref + 0.1, (name, 8 * B_ok, bad, ref)


def test_hot_linear_recurrence_alone_is_not_enough(oracle):
    """why the library cuts the batch instead of special-casing frequent features: advancing the linear weights of the frequent
    features with the bias (oracle fmo_sgd_epoch_minibatch_hot) rescues a linear model at any batch, but the factor rows of the
    same features diverge through their bilinear coupling as soon as the multipliers lag."""
    O = oracle
    e, rp, y, n = DG.criteo_shaped(30000, 5, cat_ids=2000)
    tr = O.Data(e, rp, y)
    hot = O.hot_features(tr, n, 4096, 4)

    def run(k, fn):
        m = O.Model(n, k, True, True, 0.0, 0.0, 0.001)
        if k:
            m.v[:] = O.init_values(1, n, k, 0.01)
        for _ in range(2):
            fn(m)
        return logloss(O, m, tr)
    on0 = run(0, lambda m: O.sgd_epoch_online(m, tr, 1, 0.01, -1.0, 1.0))
    hot0 = run(0, lambda m: O.sgd_epoch_minibatch(m, tr, 1, 0.01, -1.0, 1.0, 4096, 256, bias_lag=2, hot=hot))
    plain0 = run(0, lambda m: O.sgd_epoch_minibatch(m, tr, 1, 0.01, -1.0, 1.0, 4096, 256, bias_lag=2))
    assert abs(hot0 - on0) < 0.005 and not (abs(plain0 - on0) < 0.05)
    hot8 = run(8, lambda m: O.sgd_epoch_minibatch(m, tr, 1, 0.01, -1.0, 1.0, 4096, 256, bias_lag=2, hot=hot))
    on8 = run(8, lambda m: O.sgd_epoch_online(m, tr, 1, 0.01, -1.0, 1.0))
    assert not (abs(hot8 - on8) < 0.05)


def test_hot_rule_collapses_to_the_reference_at_batch_one(oracle):
    O = oracle
    e, rp, y, n = DG.criteo_shaped(300, 7, cat_ids=50)
    tr = O.Data(e, rp, y)
    a = O.Model(n, 4, True, True, 0.0, 0.001, 0.002)
    a.v[:] = O.init_values(3, n, 4, 0.05)
    b = a.copy()
    O.sgd_epoch_online(a, tr, 1, 0.05, -1.0, 1.0)
    O.sgd_epoch_minibatch(b, tr, 1, 0.05, -1.0, 1.0, 1, 1, bias_lag=1, hot=O.hot_features(tr, n, 1, 0))
    np.testing.assert_allclose(b.v, a.v, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(b.w, a.w, rtol=1e-10, atol=1e-13)
    assert abs(a.w0 - b.w0) < 1e-12


def test_two_level_rule_limits_and_stability(oracle):
    """the two-level batch rule (oracle fmo_sgd_epoch_twolevel; DESIGN.md section 9 -- restated and checked on the CPU, no device kernel
    runs it yet): hot features frozen per window, cold ones per batch.
      * limits: every feature hot == the plain rule at batch = window; no hot feature and window == batch == the plain rule at that batch;
        batch = window = chunk = 1 == the online loop;
      * stability: on Criteo-shaped rows, with the features that carry the collision mass hot (window 256) and the rest frozen for the
        WHOLE data set, the rule ends where the online loop ends -- where the plain rule at a batch of 8 192 has long diverged: the
        bound is  lr * curvature * (window * C_hot + batch * C_cold) <= 1,  C = C_hot + C_cold."""
    O = oracle
    e, rp, y, n = DG.criteo_shaped(36000, 5, cat_ids=2000)
    z = 39
    tr, te = O.Data(e[:30000 * z], rp[:30001], y[:30000]), O.Data(e[30000 * z:], rp[30000:] - rp[30000], y[30000:])
    lr = 0.01

    def fresh():
        m = O.Model(n, 8, True, True, 0.0, 0.0005, 0.001)
        m.v[:] = O.init_values(1, n, 8, 0.01)
        return m
    small = O.Data(tr.entries[:3000 * z], tr.row_ptr[:3001], tr.target[:3000])
    a, b = fresh(), fresh()
    O.sgd_epoch_twolevel(a, small, 1, lr, -1.0, 1.0, 1024, 128, 32, 2, np.ones(n, dtype=np.uint8))
    O.sgd_epoch_minibatch(b, small, 1, lr, -1.0, 1.0, 128, 32, bias_lag=2)
    np.testing.assert_allclose(a.v, b.v, rtol=1e-12, atol=1e-15); np.testing.assert_allclose(a.w, b.w, rtol=1e-12, atol=1e-15)
    assert abs(a.w0 - b.w0) < 1e-14
    a, b = fresh(), fresh()
    O.sgd_epoch_twolevel(a, small, 1, lr, -1.0, 1.0, 512, 512, 64, 1, None)
    O.sgd_epoch_minibatch(b, small, 1, lr, -1.0, 1.0, 512, 64, bias_lag=1)
    np.testing.assert_allclose(a.v, b.v, rtol=1e-12, atol=1e-15); np.testing.assert_allclose(a.w, b.w, rtol=1e-12, atol=1e-15)
    tiny = O.Data(tr.entries[:200 * z], tr.row_ptr[:201], tr.target[:200])
    a, b = fresh(), fresh()
    O.sgd_epoch_twolevel(a, tiny, 1, lr, -1.0, 1.0, 1, 1, 1, 0, None)
    O.sgd_epoch_online(b, tiny, 1, lr, -1.0, 1.0)
    np.testing.assert_allclose(a.v, b.v, rtol=1e-10, atol=1e-14); np.testing.assert_allclose(a.w, b.w, rtol=1e-10, atol=1e-14)
    # stability: hot = the most frequent features until the rest's share of C, times the WHOLE data set, stays below 1/2
    cnt = np.bincount(tr.entries["id"], minlength=n).astype(np.float64)
    p2 = (cnt / tr.n_rows) ** 2
    order = np.argsort(-p2)
    tail = p2.sum() - np.cumsum(p2[order])
    n_hot = int(np.argmax(lr * 0.25 * tr.n_rows * tail <= 0.5)) + 1
    hot = np.zeros(n, dtype=np.uint8)
    hot[order[:n_hot]] = 1
    C_hot = float(p2[order[:n_hot]].sum())
    assert n_hot < 0.2 * n and lr * 0.25 * 256 * C_hot <= 1.0

    def run(step):
        m = fresh()
        for _ in range(3):
            step(m)
        return logloss(O, m, te)
    ref = run(lambda m: O.sgd_epoch_online(m, tr, 1, lr, -1.0, 1.0))
    two = run(lambda m: O.sgd_epoch_twolevel(m, tr, 1, lr, -1.0, 1.0, tr.n_rows, 256, 256, 1, hot))
    plain = run(lambda m: O.sgd_epoch_minibatch(m, tr, 1, lr, -1.0, 1.0, 8192, 256, bias_lag=2))
    assert abs(two - ref) <= 0.02, (two, ref)
    assert not np.isfinite(plain) or plain > ref + 0.1, (plain, ref)
