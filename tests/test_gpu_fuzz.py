"""GPU: seeded random shapes through every form of the batch rule (one pass, split step, dense owner pass; SGD and the lambda-
step learner) against the oracle's rule -- ragged rows with empty ones and repeated ids, Zipf ids (most features collide),
factor counts on both sides of every padding boundary, batches that divide nothing, micro-chunks of 1 .. 300, lags 1 .. 4.
1e-4 relative like every parity test.  The point is the combinations nobody wrote a named case for."""
import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu


def _seeds(default):
    """the seeds of a fuzz test: range(default), or FMX_FUZZ_SEEDS=lo:hi for a soak run over other seeds (scripts/gpu_fuzz_soak.sh)"""
    import os
    v = os.environ.get("FMX_FUZZ_SEEDS")
    if not v:
        return range(default)
    lo, hi = (int(x) for x in v.split(":"))
    return range(lo, hi)


def _split(monkeypatch, v):
    """fmx_config::als_split_min for the handles created from here on: "0" = never split (fused draws), "1" = every level, n = levels of >= n entries"""
    from libfm_amd import capi as _c
    monkeypatch.setattr(_c, "ALS_SPLIT_MIN", _c.ALS_SPLIT_NEVER if str(v) == "0" else int(v))

RTOL = 1e-4


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import build, capi
    build.build()
    if capi.load().fmx_device_count() == 0:
        pytest.fail("gpu-marked test without a HIP device")
    return capi


def _floor(run):
    """What no fp32 implementation of a rule can be held below on a given case: the oracle's own response to a change of the initial
    factors by ONE fp32 rounding (6e-8 relative).  On well-conditioned cases it is far below the base tolerances; random step sizes on
    random data also produce iterations that amplify a rounding a million-fold or leave the numbers altogether (soak runs over other
    seeds, scripts/gpu_fuzz_soak.sh, hit a few per thousand) -- those are compared at eight times their own floor, or skipped.
    run(pert) -> tuple of arrays / scalars; returns (floors | None, run(0))."""
    a, b = run(0.0), run(6e-8)
    out = []
    for x, y in zip(a, b):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        if not (np.isfinite(x).all() and np.isfinite(y).all()):
            return None, a
        out.append(float(np.abs(x - y).max()) if x.size else 0.0)
    return out, a


def _case(seed, als=False):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31, 32, 33, 64, 65, 100, 128, 129]))
    task = int(rng.integers(0, 2))
    kind = int(rng.integers(0, 3))
    rows = int(rng.integers(50, 900))
    if kind == 0:                                              # ragged real-valued rows, empty rows, a repeated id per row
        n = int(rng.integers(40, 400))
        # (ALS: no repeated ids -- the reference's closed-form coordinate step is not a minimiser then and its own fp64 sweep
        #  blows up on such data, seed 108: 1e306 after two iterations)
        data = datagen.ragged_real(n, rows, int(rng.integers(2, 70)), seed, classification=bool(task),
                                   empty_every=int(rng.choice([0, 5, 11])), duplicates=bool(rng.integers(0, 2)) and not als)
    else:                                                      # one-hot fields, uniform or Zipf ids
        nnz = int(rng.choice([3, 8, 12, 24, 33, 40]))
        n = nnz * int(rng.integers(5, 200))
        data = datagen.onehot_fields(n, nnz, rows, seed, zipf=float(rng.choice([0.0, 1.1])), classification=bool(task))
    batch = int(rng.choice([1, 7, 33, 64, 100, 256, 257, 1000]))
    chunk = int(rng.choice([1, 3, 16, 50, 256, 300]))
    lag = int(rng.integers(1, 5))
    k0, k1 = bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 4) > 0)
    if als and kind == 0 and task == 0:
        k1 = True          # real-valued rows, targets of +-25 and NO linear term: the sweep amplifies the fp32 rounding of the stored
        #                    parameters factor after factor (seed 103: 1e-6, 5e-6, ... 5e-4 at the 14th factor; 1e-7 with the linear term)
    return n, k, task, data, batch, chunk, lag, k0, k1


@pytest.mark.parametrize("seed", _seeds(24))
def test_random_shape_every_form_of_the_rule(capi, oracle, seed):
    n, k, task, (ent, rp, y), batch, chunk, lag, k0, k1 = _case(seed)
    lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
    lr = 0.003
    d = oracle.Data(ent, rp, y)
    forms = [(capi.APPLY_FUSED, 0, lag), (capi.APPLY_DEFAULT, capi.FLAG_BIAS_LAG, lag), (capi.APPLY_SEGMENTED, capi.FLAG_BIAS_LAG, lag),
             (capi.APPLY_DEFAULT, 0, 0)]                       # the last one: bias recurrence coupled exactly (no lag)
    for apply_, flags, lg in forms:
        def run(pert):
            m = oracle.Model(n, k, k0, k1, 0.001 if k0 else 0.0, 0.002, 0.004)
            m.v[:] = oracle.init_values(21 + seed, n, k, 0.05) * (1.0 + pert)
            if k1:
                m.w[:] = oracle.init_values(22 + seed, n, 1, 0.05)[0]
            m.w0 = 0.02 if k0 else 0.0
            for _ in range(2):
                oracle.sgd_epoch_minibatch(m, d, task, lr, lo, hi, batch, chunk, bias_lag=lg)
            return m.w0, m.w.copy(), m.v.copy(), oracle.predict_raw(m, d)
        what = "seed %d form (%d,%d,%d): n=%d k=%d batch=%d chunk=%d" % (seed, apply_, flags, lg, n, k, batch, chunk)
        fl, (o_w0, o_w, o_v, o_p) = _floor(run)
        if fl is None or fl[2] > 1e-3:
            assert seed >= 24, what + ": a case of the suite's own seeds must be well-conditioned"
            continue                                            # the iteration amplifies roundings beyond comparison (soak seeds only)
        h = capi.Handle(n, k, k0, k1, task, 0.001 if k0 else 0.0, 0.002, 0.004, lr, lo, hi)
        h.set_params(0.02 if k0 else 0.0, oracle.init_values(22 + seed, n, 1, 0.05)[0] if k1 else np.zeros(n), oracle.init_values(21 + seed, n, k, 0.05))
        h.upload_rows(0, ent, rp, y)
        for _ in range(2):
            h.sgd_epoch(0, capi.SGD_MINIBATCH, apply_, batch, chunk, flags, lg)
        w0, w, v = h.get_params()
        assert abs(w0 - o_w0) <= RTOL * abs(o_w0) + 1e-5 + 8 * fl[0], what
        np.testing.assert_allclose(w, o_w, rtol=RTOL, atol=2e-5 + 8 * fl[1], err_msg=what)
        np.testing.assert_allclose(v, o_v, rtol=RTOL, atol=2e-5 + 8 * fl[2], err_msg=what)
        np.testing.assert_allclose(h.predict(0, d.n_rows), o_p, rtol=RTOL, atol=5e-5 + 8 * fl[3], err_msg=what)
        h.close()


@pytest.mark.parametrize("seed", _seeds(12))
def test_random_shape_als_both_draw_forms(capi, oracle, seed, monkeypatch):
    """fm_learn_mcmc without sampling on random shapes: fused draws and the split step (forced for every level) against the oracle."""
    from libfm_amd import learner as L
    import io
    n, k, task, (ent, rp, y), _, _, _, k0, k1 = _case(100 + seed, als=True)
    k = min(k, 33)                                             # (the oracle's sweep is O(k nnz) per iteration: keep the CPU side short)
    rng = np.random.default_rng(77 + seed)
    n_test = 60
    lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
    tr = (ent, rp, y)
    te_rows = slice(0, n_test)
    te_rp = rp[:n_test + 1].copy()
    te = (ent[:int(te_rp[-1])].copy(), te_rp, y[:n_test].copy())
    wl, vl = float(rng.uniform(0.5, 3.0)), float(rng.uniform(1.0, 10.0))
    ref = None
    for split_min in ("0", "1"):
        _split(monkeypatch, split_min)
        m = oracle.Model(n, k, k0, k1, 0.0, wl, vl)
        m.v[:] = oracle.init_values(31 + seed, n, k, 0.1)
        if k1:
            m.w[:] = oracle.init_values(32 + seed, n, 1, 0.1)[0]
        m.w0 = 0.0
        fm = L.FMModel()
        fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = n, k, k0, k1
        fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
        l = L.FMLearnALS()
        l.fm, l.task, l.num_iter, l.min_target, l.max_target, l.w_lambda, l.v_lambda = fm, task, 3, lo, hi, wl, vl
        l.out = io.StringIO()
        l.init()
        l.learn(L.Data(*tr), L.Data(*te))
        if ref is None:
            pred, _ = oracle.als_learn(m, oracle.Data(*tr), oracle.Data(*te), task, 3, wl, vl, lo, hi)
            ref = (m.w0, m.w.copy(), m.v.copy(), pred)
        what = "seed %d split_min %s: n=%d k=%d task=%d k0=%d k1=%d" % (seed, split_min, n, k, task, k0, k1)
        assert abs(l.fm.w0 - ref[0]) <= RTOL * abs(ref[0]) + 2e-5, what
        np.testing.assert_allclose(l.fm.w, ref[1], rtol=RTOL, atol=2e-5, err_msg=what)
        np.testing.assert_allclose(l.fm.v, ref[2], rtol=RTOL, atol=2e-5, err_msg=what)
        np.testing.assert_allclose(l.pred_this, ref[3], rtol=RTOL, atol=5e-5, err_msg=what)
        l.close()


@pytest.mark.parametrize("seed", _seeds(8))
def test_random_shape_feature_shards(capi, oracle, seed):
    """2 .. 5 loopback shards (hashed or plain ownership, exact or pipelined schedule) on random shapes against the oracle's rule."""
    n, k, task, (ent, rp, y), batch, chunk, lag, k0, k1 = _case(200 + seed)
    rng = np.random.default_rng(300 + seed)
    world, shard_hash, pipeline = int(rng.integers(2, 6)), int(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0)
    lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
    lr = 0.003
    d = oracle.Data(ent, rp, y)
    w_init = oracle.init_values(42 + seed, n, 1, 0.05)[0] if k1 else np.zeros(n)

    def run(pert):
        m = oracle.Model(n, k, k0, k1, 0.001 if k0 else 0.0, 0.002, 0.004)
        m.v[:] = oracle.init_values(41 + seed, n, k, 0.05) * (1.0 + pert)
        m.w[:] = w_init
        m.w0 = 0.02 if k0 else 0.0
        for _ in range(2):
            oracle.sgd_epoch_minibatch(m, d, task, lr, lo, hi, batch, chunk, bias_lag=lag, pipelined=pipeline)
        return m.w0, m.w.copy(), m.v.copy(), oracle.predict_raw(m, d)
    what = "seed %d: world=%d hash=%d pipeline=%d n=%d k=%d batch=%d chunk=%d lag=%d" % (seed, world, shard_hash, pipeline, n, k, batch, chunk, lag)
    fl, (o_w0, o_w, o_v, o_p) = _floor(run)
    if fl is None or fl[2] > 1e-3:
        assert seed >= 8, what + ": a case of the suite's own seeds must be well-conditioned"
        pytest.skip("the iteration amplifies roundings beyond comparison")
    hs = [capi.Handle(n, k, k0, k1, task, 0.001 if k0 else 0.0, 0.002, 0.004, lr, lo, hi, device=0, shard_rank=r, shard_world=world,
                      shard_hash=shard_hash) for r in range(world)]
    for h in hs:
        h.set_params(0.02 if k0 else 0.0, w_init, oracle.init_values(41 + seed, n, k, 0.05))
        h.upload_rows(0, ent, rp, y)
    grp = capi.Group(hs)
    flags = capi.FLAG_BIAS_LAG | (capi.FLAG_PIPELINE if pipeline else 0)
    for _ in range(2):
        grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, batch, chunk, flags, lag)
    w0, w, v = grp.get_params()
    assert abs(w0 - o_w0) <= RTOL * abs(o_w0) + 1e-5 + 8 * fl[0], what
    np.testing.assert_allclose(w, o_w, rtol=RTOL, atol=2e-5 + 8 * fl[1], err_msg=what)
    np.testing.assert_allclose(v, o_v, rtol=RTOL, atol=2e-5 + 8 * fl[2], err_msg=what)
    np.testing.assert_allclose(grp.predict(0, d.n_rows), o_p, rtol=RTOL, atol=5e-5 + 8 * fl[3], err_msg=what)
    grp.close()
    for h in hs:
        h.close()


@pytest.mark.parametrize("seed", _seeds(10))
def test_random_shape_sgda_both_forms(capi, oracle, seed):
    """fm_learn_sgd_element_adapt_reg on random shapes (odd factor counts, 1 .. 4 attribute groups): the device learner in
    reference order against the restated online loop, the batch form against its restated rule."""
    n, k, task, (ent, rp, y), batch, chunk, _, _, _ = _case(400 + seed)
    k = min(k, 65)
    rng = np.random.default_rng(500 + seed)
    G = int(rng.integers(1, 5))
    group = None if G == 1 else rng.integers(0, G, n).astype(np.uint32)
    if group is not None:
        group[:G] = np.arange(G)                               # every group is used
    lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
    lr = 0.002
    rows = len(y)
    n_val = max(20, rows // 3)
    va_rp = rp[:n_val + 1].copy()
    tr, va = oracle.Data(ent, rp, y), oracle.Data(ent[:int(va_rp[-1])].copy(), va_rp, y[:n_val].copy())
    for b in (None, batch):                                    # None: the reference's online order
        def run(pert):
            m = oracle.Model(n, k, True, True, 0.0, 0.0, 0.0)
            m.v[:] = oracle.init_values(51 + seed, n, k, 0.05) * (1.0 + pert)
            st = oracle.sgda_learn(m, tr, va, task, lr, lo, hi, 3, group, batch=b, w0_chunk=chunk)
            return m.w0, m.w.copy(), m.v.copy(), np.array(st.reg_w, dtype=np.float64), np.array(st.reg_v, dtype=np.float64)[:, :k]
        what = "seed %d batch %s: n=%d k=%d task=%d G=%d chunk=%d" % (seed, b, n, k, task, G, chunk)
        fl, (o_w0, o_w, o_v, o_rw, o_rv) = _floor(run)
        if fl is None or fl[2] > 1e-3:
            assert seed >= 10, what + ": a case of the suite's own seeds must be well-conditioned"
            continue                                            # the iteration amplifies roundings beyond comparison (soak seeds only)
        h = capi.Handle(n, k, True, True, task, 0.0, 0.0, 0.0, lr, lo, hi)
        h.set_params(0.0, np.zeros(n), oracle.init_values(51 + seed, n, k, 0.05))
        if group is not None:
            h.set_groups(group)
        h.upload_rows(0, tr.entries, tr.row_ptr, tr.target)
        h.upload_rows(1, va.entries, va.row_ptr, va.target)
        h.sgda_begin()
        for i in range(3):
            if b is None:
                h.sgda_epoch(0, 1, i > 0)
            else:
                h.sgda_epoch_minibatch(0, 1, i > 0, b, chunk)
        reg = h.sgda_get_reg()
        w0, w, v = h.get_params()
        h.sgda_end()
        h.close()
        assert abs(w0 - o_w0) <= RTOL * abs(o_w0) + 2e-5 + 8 * fl[0], what
        np.testing.assert_allclose(w, o_w, rtol=RTOL, atol=2e-5 + 8 * fl[1], err_msg=what)
        np.testing.assert_allclose(v, o_v, rtol=RTOL, atol=2e-5 + 8 * fl[2], err_msg=what)
        np.testing.assert_allclose(reg[:, 0], o_rw, rtol=1e-3, atol=1e-7 + 8 * fl[3], err_msg=what)
        np.testing.assert_allclose(reg[:, 1:1 + k], o_rv, rtol=1e-3, atol=1e-7 + 8 * fl[4], err_msg=what)


@pytest.mark.parametrize("k,seed", [(1, 0), (5, 1), (33, 2), (70, 3)])
def test_kept_blocks_equal_joined_rows_at_odd_factor_counts(capi, oracle, k, seed, monkeypatch):
    """`-relation` blocks kept apart (per-block-row caches) against the device join, ALS and a sampled chain, factor counts that
    are not their own padding; main features in the split and in the fused form of the draws."""
    from libfm_amd import data as D
    from libfm_amd import learner as L
    import io
    (ent, rp, y), blocks, maps = datagen.block_structured(30, 20, 250, seed=60 + seed)
    n_main = 7
    _, _, offs = datagen.expand_blocks(ent, rp, blocks, maps, n_main)
    n = offs[-1] + blocks[-1][2]
    res = []
    for keep, split, sample in ((True, "0", False), (False, "0", False), (True, "1", False), (True, "0", True), (False, "1", True)):
        _split(monkeypatch, split)
        fm = L.FMModel()
        fm.num_attribute, fm.num_factor = n, k
        fm.w0, fm.w, fm.v = 0.0, oracle.init_values(71, n, 1, 0.1)[0].copy(), oracle.init_values(72, n, k, 0.1).copy()
        l = L.FMLearnALS()
        l.fm, l.task, l.num_iter, l.min_target, l.max_target = fm, 0, 3, float(y.min()), float(y.max())
        l.w_lambda, l.v_lambda, l.do_sample, l.seed = 1.5, 4.0, sample, 9
        l.out = io.StringIO()
        train = L.Data(ent, rp, y)
        for (be, bp, nf), mp, off in zip(blocks, maps, offs):
            train.add_relation(D.Relation(be, bp, nf), mp, off)
        train.keep_blocks = keep
        l.init()
        l.learn(train, train)
        res.append((sample, l.fm.w0, l.fm.w.copy(), l.fm.v.copy()))
        l.close()
    for sample in (False, True):
        same = [r for r in res if r[0] == sample]
        for other in same[1:]:
            assert abs(other[1] - same[0][1]) <= RTOL * abs(same[0][1]) + 2e-5
            np.testing.assert_allclose(other[2], same[0][2], rtol=RTOL, atol=5e-5)
            np.testing.assert_allclose(other[3], same[0][3], rtol=RTOL, atol=5e-5)


@pytest.mark.parametrize("seed", range(6))
def test_random_shape_sharded_als(capi, oracle, seed):
    """fm_learn_mcmc over 2 .. 4 feature shards (global dependency levels, one all-reduce per (family, level)) on random shapes:
    ALS against the oracle, and a sampled chain against the chain one handle draws with the same seed."""
    n, k, task, (ent, rp, y), _, _, _, k0, k1 = _case(600 + seed, als=True)
    k = min(k, 33)
    rng = np.random.default_rng(700 + seed)
    world, shard_hash = int(rng.integers(2, 5)), int(rng.integers(0, 2))
    lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
    wl, vl = float(rng.uniform(0.5, 3.0)), float(rng.uniform(1.0, 10.0))
    d = oracle.Data(ent, rp, y)
    init_v, init_w = oracle.init_values(81 + seed, n, k, 0.1), oracle.init_values(82 + seed, n, 1, 0.1)[0]

    def run(world, do_sample):
        hs = [capi.Handle(n, k, k0, k1, task, 0.0, wl, vl, 0.0, lo, hi, device=0, shard_rank=r, shard_world=world,
                          shard_hash=shard_hash) for r in range(world)]
        for h in hs:
            h.set_params(0.0, init_w if k1 else np.zeros(n), init_v)
            h.upload_rows(0, ent, rp, y)
        grp = capi.Group(hs) if world > 1 else None
        drv = grp if grp else hs[0]
        drv.als_begin(0)
        for _ in range(3):
            drv.als_sweep(wl, vl, do_sample=do_sample, seed=5)
        drv.als_end()
        out = grp.get_params() if grp else hs[0].get_params()
        if grp:
            grp.close()
        for h in hs:
            h.close()
        return out

    m = oracle.Model(n, k, k0, k1, 0.0, wl, vl)
    m.v[:] = init_v
    if k1:
        m.w[:] = init_w
    te = oracle.Data(ent[:int(rp[20])].copy(), rp[:21].copy(), y[:20].copy())
    oracle.als_learn(m, d, te, task, 3, wl, vl, lo, hi)
    what = "seed %d: world=%d hash=%d n=%d k=%d task=%d k0=%d k1=%d" % (seed, world, shard_hash, n, k, task, k0, k1)
    w0, w, v = run(world, False)
    assert abs(w0 - m.w0) <= RTOL * abs(m.w0) + 2e-5, what
    np.testing.assert_allclose(w, m.w, rtol=RTOL, atol=2e-5, err_msg=what)
    np.testing.assert_allclose(v, m.v, rtol=RTOL, atol=2e-5, err_msg=what)
    one, many = run(1, True), run(world, True)
    assert abs(one[0] - many[0]) <= RTOL * abs(one[0]) + 2e-5, what
    np.testing.assert_allclose(many[1], one[1], rtol=RTOL, atol=5e-5, err_msg=what)
    np.testing.assert_allclose(many[2], one[2], rtol=RTOL, atol=5e-5, err_msg=what)


@pytest.mark.parametrize("k,k0,k1", [(5, True, True), (33, False, True), (70, True, False), (0, True, True)])
def test_model_file_at_odd_factor_counts(capi, oracle, k, k0, k1, tmp_path):
    """fm_model::saveModel / loadModel through the device table when the row is padded (k < KP): the file is the Python writer's
    byte for byte (that one is the stock binary's, tests/test_data_formats.py), and loads back exactly -- one handle and 3 shards."""
    from libfm_amd import learner as L
    n = 137
    w = oracle.init_values(91 + k, n, 1, 0.3)[0] if k1 else np.zeros(n)
    v = oracle.init_values(92 + k, n, max(k, 1), 0.3)[:k]
    w32, v32 = w.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
    h = capi.Handle(n, k, k0, k1, 0, 0, 0, 0, 0.01, -1.0, 1.0)
    h.set_params(0.375 if k0 else 0.0, w, v if k else None)
    dev_file, py_file = str(tmp_path / "dev.model"), str(tmp_path / "py.model")
    h.save_model(dev_file)
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = n, k, k0, k1
    fm.w0, fm.w, fm.v = (0.375 if k0 else 0.0), w32, v32
    fm.save_model(py_file)
    assert open(dev_file).read() == open(py_file).read()
    h.close()
    got_w, got_v = np.zeros(n), np.zeros((k, n))
    for r in range(3):
        hs = capi.Handle(n, k, k0, k1, 0, 0, 0, 0, 0.01, -1.0, 1.0, shard_rank=r, shard_world=3, shard_hash=1)
        hs.load_model(py_file)
        w0, got_w, got_v = hs.get_params(got_w, got_v)
        assert w0 == (0.375 if k0 else 0.0)
        hs.close()
    fm2 = L.FMModel()                                          # (the text file holds 6 significant digits: compare with its reader)
    fm2.num_attribute, fm2.num_factor, fm2.k0, fm2.k1 = n, k, k0, k1
    assert fm2.load_model(py_file)
    want_w = fm2.w.astype(np.float32).astype(np.float64) if k1 else np.zeros(n)
    assert np.array_equal(got_w, want_w) and np.array_equal(got_v, fm2.v.astype(np.float32).astype(np.float64)[:k])


def test_no_factors_at_all(capi, oracle, monkeypatch):
    """`-dim 1,1,0` (a linear model: legal in libFM) through ALS in both draw forms, both SGDA forms, sharded SGD and sharded ALS"""
    n, nnz, rows = 300, 6, 400
    ent, rp, y = datagen.onehot_fields(n, nnz, rows, seed=5, zipf=1.05, classification=False)
    d = oracle.Data(ent, rp, y)
    lo, hi = float(y.min()), float(y.max())
    te = oracle.Data(ent[:int(rp[20])].copy(), rp[:21].copy(), y[:20].copy())
    w_init = oracle.init_values(3, n, 1, 0.1)[0]

    def fresh():
        m = oracle.Model(n, 0, True, True, 0.0, 1.0, 1.0)
        m.w[:] = w_init
        return m
    # ALS, one handle (fused / split draws) and 3 shards
    ref = fresh()
    oracle.als_learn(ref, d, te, 0, 3, 1.0, 1.0, lo, hi)
    for world, split in ((1, "0"), (1, "1"), (3, "0")):
        _split(monkeypatch, split)
        hs = [capi.Handle(n, 0, True, True, 0, 0.0, 1.0, 1.0, 0.0, lo, hi, device=0, shard_rank=r, shard_world=world, shard_hash=1)
              for r in range(world)]
        for h in hs:
            h.set_params(0.0, w_init, None)
            h.upload_rows(0, ent, rp, y)
        grp = capi.Group(hs) if world > 1 else None
        drv = grp if grp else hs[0]
        drv.als_begin(0)
        for _ in range(3):
            drv.als_sweep(1.0, 1.0)
        drv.als_end()
        w0, w, _ = grp.get_params() if grp else hs[0].get_params()
        assert abs(w0 - ref.w0) <= RTOL * abs(ref.w0) + 2e-5
        np.testing.assert_allclose(w, ref.w, rtol=RTOL, atol=2e-5)
        if grp:
            grp.close()
        for h in hs:
            h.close()
    # SGD over 2 shards, SGDA in both forms
    lr = 0.01
    m = fresh()
    hs = [capi.Handle(n, 0, True, True, 0, 0.0, 0.001, 0.0, lr, lo, hi, device=0, shard_rank=r, shard_world=2) for r in range(2)]
    for h in hs:
        h.set_params(0.0, w_init, None)
        h.upload_rows(0, ent, rp, y)
    grp = capi.Group(hs)
    m.regw = 0.001
    for _ in range(2):
        grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, 64, 8, capi.FLAG_BIAS_LAG, 2)
        oracle.sgd_epoch_minibatch(m, d, 0, lr, lo, hi, 64, 8, bias_lag=2)
    w0, w, _ = grp.get_params()
    assert abs(w0 - m.w0) <= RTOL * abs(m.w0) + 1e-5
    np.testing.assert_allclose(w, m.w, rtol=RTOL, atol=2e-5)
    grp.close()
    for h in hs:
        h.close()
    for b in (None, 50):
        m = fresh()
        m.regw = 0.0
        h = capi.Handle(n, 0, True, True, 0, 0.0, 0.0, 0.0, 0.002, lo, hi)
        h.set_params(0.0, w_init, None)
        h.upload_rows(0, ent, rp, y)
        h.upload_rows(1, te.entries, te.row_ptr, te.target)
        h.sgda_begin()
        for i in range(3):
            h.sgda_epoch(0, 1, i > 0) if b is None else h.sgda_epoch_minibatch(0, 1, i > 0, b, 4)
        reg = h.sgda_get_reg()
        w0, w, _ = h.get_params()
        h.sgda_end()
        h.close()
        st = oracle.sgda_learn(m, d, te, 0, 0.002, lo, hi, 3, None, batch=b, w0_chunk=4)
        assert abs(w0 - m.w0) <= RTOL * abs(m.w0) + 2e-5
        np.testing.assert_allclose(w, m.w, rtol=RTOL, atol=2e-5)
        np.testing.assert_allclose(reg[:, 0], st.reg_w, rtol=1e-3, atol=1e-7)


def test_empty_and_single_row_data_sets(capi, oracle):
    """0 rows and 1 row through upload / predict / evaluate / every SGD form / SGDA / ALS: no crash, the one-row results are the rule's"""
    n, k = 50, 8
    empty = (np.zeros(0, dtype=datagen.ENTRY_DTYPE), np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.float32))
    h = capi.Handle(n, k, True, True, 0, 0.0, 0.001, 0.001, 0.01, -2.0, 2.0)
    h.init_params(0.1, 0.1, 7)
    w0_before = h.get_w0()
    h.upload_rows(0, *empty)
    assert len(h.predict(0, 0)) == 0
    assert h.evaluate(0).rows == 0
    for mode, apply_, flags, lag in ((capi.SGD_SEQUENTIAL, capi.APPLY_DEFAULT, 0, 0), (capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, 0, 0),
                                     (capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 2), (capi.SGD_HOGWILD, capi.APPLY_STORE, 0, 0)):
        st = h.sgd_epoch(0, mode, apply_, 16, 4, flags, lag)
        assert st.rows == 0
    assert h.get_w0() == w0_before
    with pytest.raises(capi.FmxError):
        h.als_begin(0)                                         # "empty training set"
    # one row
    ent = np.zeros(3, dtype=datagen.ENTRY_DTYPE)
    ent["id"], ent["value"] = [4, 9, 30], [1.0, -0.5, 2.0]
    rp, y = np.array([0, 3], dtype=np.uint64), np.array([1.5], dtype=np.float32)
    d = oracle.Data(ent, rp, y)
    w0, w, v = h.get_params()
    for apply_, flags, lag in ((capi.APPLY_FUSED, 0, 1), (capi.APPLY_DEFAULT, 0, 0), (capi.APPLY_SEGMENTED, capi.FLAG_BIAS_LAG, 2)):
        m = oracle.Model(n, k, True, True, 0.0, 0.001, 0.001)
        m.w0, m.w[:], m.v[:] = w0, w, v
        h.set_params(w0, w, v)
        h.upload_rows(1, ent, rp, y)
        h.sgd_epoch(1, capi.SGD_MINIBATCH, apply_, 16, 4, flags, lag)
        oracle.sgd_epoch_minibatch(m, d, 0, 0.01, -2.0, 2.0, 16, 4, bias_lag=lag)
        g0, gw, gv = h.get_params()
        assert abs(g0 - m.w0) <= RTOL * abs(m.w0) + 1e-6
        np.testing.assert_allclose(gw, m.w, rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(gv, m.v, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(h.predict(1, 1), oracle.predict_raw(m, d), rtol=RTOL, atol=1e-5)
    h.als_begin(1)
    h.als_sweep(1.0, 2.0)
    h.als_end()
    h.close()


@pytest.mark.parametrize("regime", ["stable", "marginal"])
@pytest.mark.parametrize("seed", _seeds(8))
def test_random_shape_large_batches_hand_off_and_side_stream(capi, oracle, seed, regime):
    """the one-pass form where a batch holds >= 32 768 rows (the recurrence on the side stream): random rows / batch / micro-chunk / lag,
    the two orderings of the streams (device-side hand-off, events) and the weight side stream kept or not, at random -- parameters and
    the predictions of the pass that follows (out of the side stream when it was kept) against the oracle's rule."""
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.choice([4, 8, 16, 33, 64]))
    task = int(rng.integers(0, 2))
    rows = int(rng.integers(66000, 120000))
    if rng.integers(0, 2):
        nnz = int(rng.choice([4, 9, 16]))
        n = nnz * int(rng.integers(2000, 40000))
        ent, rp, y = datagen.onehot_fields(n, nnz, rows, seed, zipf=float(rng.choice([0.0, 0.9])), classification=bool(task))
    else:
        n = int(rng.integers(20000, 200000))
        ent, rp, y = datagen.ragged_real(n, rows, int(rng.integers(3, 20)), seed, classification=bool(task), empty_every=int(rng.choice([0, 13])))
    batch = int(rng.choice([32768, 33001, 50000, 65536]))
    chunk = int(rng.choice([64, 256, 300, 512, 32, 1]))          # (powers of two: the parallel-in-time recurrence; 300: the one-wavefront chain)
    lag = int(rng.integers(1, 5))
    flags = (capi.FLAG_EVENT_SYNC if rng.integers(0, 2) else 0) | (capi.FLAG_KEEP_WSIDE if rng.integers(0, 2) else 0)
    lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
    curv = 1.0 if task == 0 else 0.25
    # a step size the batch rule is STABLE at on these rows (lr * curvature * batch * C <= 1/2): beyond it the iteration
    # amplifies every fp32 rounding and no implementation can be held to the fp64 oracle (soak seeds 233 / 248: Zipf ids, batch 33 001 --
    # one-pass and two-pass forms alike 1e-2 off, side stream and hand-off identical to the plain forms)
    # "marginal" (round-4 verdict, housekeeping): the region between gain 1/2 and 2 that the cap above had left unprobed -- the rule degrades
    # there but is defined, and the device must still follow the fp64 oracle to within the oracle's OWN response to one fp32 rounding of the
    # start values (the floor below; a case whose floor has left 1e-3 is not a statement about any implementation and is skipped)
    C = datagen.collision_mass(ent, rows, n)
    gain = 0.5 if regime == "stable" else float(rng.uniform(0.6, 1.8))
    lr = min(0.004 if regime == "stable" else 0.02, 0.9 / (chunk * curv), gain / (curv * batch * max(C, 1e-12)))
    d = oracle.Data(ent, rp, y)
    what = "seed %d (%s): n=%d k=%d rows=%d batch=%d chunk=%d lag=%d flags=%d lr=%.3g gain=%.2f" % (seed, regime, n, k, rows, batch, chunk, lag, flags, lr, lr * curv * batch * C)

    def run(pert):
        m = oracle.Model(n, k, True, True, 0.001, 0.002, 0.004)
        m.v[:] = oracle.init_values(31 + seed, n, k, 0.05) * (1.0 + pert)
        m.w[:] = oracle.init_values(32 + seed, n, 1, 0.05)[0]
        m.w0 = 0.02
        for _ in range(2):
            oracle.sgd_epoch_minibatch(m, d, task, lr, lo, hi, batch, chunk, bias_lag=lag)
        return m.w0, m.w.copy(), m.v.copy(), oracle.predict_raw(m, d)
    fl, (o_w0, o_w, o_v, o_p) = _floor(run)
    if fl is None or fl[2] > 1e-3:
        assert seed >= 8 or regime == "marginal", what + ": a stable case of the suite's own seeds must be well-conditioned"
        pytest.skip(what + ": the oracle itself amplifies one fp32 rounding beyond 1e-3 here")
    h = capi.Handle(n, k, True, True, task, 0.001, 0.002, 0.004, lr, lo, hi)
    h.set_params(0.02, oracle.init_values(32 + seed, n, 1, 0.05)[0], oracle.init_values(31 + seed, n, k, 0.05))
    h.upload_rows(0, ent, rp, y)
    for _ in range(2):
        h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, batch, chunk, flags, lag)
    assert bool(h.evaluate(0).flags & capi.EVAL_WSIDE) == bool(flags & capi.FLAG_KEEP_WSIDE), what
    np.testing.assert_allclose(h.predict(0, d.n_rows), o_p, rtol=RTOL, atol=5e-5 + 8 * fl[3], err_msg=what)
    w0, w, v = h.get_params()
    assert abs(w0 - o_w0) <= RTOL * abs(o_w0) + 1e-5 + 8 * fl[0], what
    np.testing.assert_allclose(w, o_w, rtol=RTOL, atol=2e-5 + 8 * fl[1], err_msg=what)
    np.testing.assert_allclose(v, o_v, rtol=RTOL, atol=2e-5 + 8 * fl[2], err_msg=what)
    h.close()
