"""GPU: several feature shards driven through the C-ABI alone (fmx_group_*; no torch, no Python in the step).
On a one-GPU box the shards share the device and the exchange is the library's loopback reduction kernel; with two or
more visible devices the same calls run over RCCL.  Bar: the sharded run trains the SAME model as one unsharded handle
under the same rule (batch, micro-chunk, bias lag) -- 1e-4 -- and both sit on the oracle's rule."""
import numpy as np
import pytest

import datagen
from common import Golden

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import build, capi
    build.build()
    if capi.load().fmx_device_count() == 0:
        pytest.fail("gpu-marked test without a HIP device")
    return capi


def make_group(capi, world, devices, n, k, task, reg, lr, lo, hi, shard_hash, k0=True, k1=True):
    hs = [capi.Handle(n, k, k0, k1, task, reg[0], reg[1], reg[2], lr, lo, hi, device=devices[r], shard_rank=r,
                      shard_world=world, shard_hash=shard_hash) for r in range(world)]
    return hs, capi.Group(hs)


@pytest.mark.parametrize("world,shard_hash,lag,pipeline", [(2, 0, 1, False), (2, 1, 1, False), (3, 1, 2, False), (4, 1, 1, True),
                                                           (2, 1, 3, False), (8, 1, 2, False)])
@pytest.mark.parametrize("name", ["sgd_cls_zipf_k32", "sgd_reg_ml"])
def test_group_trains_the_unsharded_model(capi, oracle, name, world, shard_hash, lag, pipeline):
    g = Golden(name)
    m = g.model(oracle, "init")
    tr, te = g.data(oracle, "train"), g.data(oracle, "test")
    batch, chunk = 100, 10
    hs, grp = make_group(capi, world, [0] * world, g.n, g.k, g.task, g.reg, g.lr, g.min_target, g.max_target, shard_hash, g.k0, g.k1)
    for h in hs:
        h.set_params(m.w0, m.w, m.v)
        h.upload_rows(0, tr.entries, tr.row_ptr, tr.target)
        h.upload_rows(1, te.entries, te.row_ptr, te.target)
    flags = capi.FLAG_BIAS_LAG | (capi.FLAG_PIPELINE if pipeline else 0)
    for _ in range(g.iters):
        grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, batch, chunk, flags, lag)
        oracle.sgd_epoch_minibatch(m, tr, g.task, g.lr, g.min_target, g.max_target, batch, chunk, bias_lag=lag, pipelined=pipeline)
    w0, w, v = grp.get_params()
    assert abs(w0 - m.w0) <= RTOL * abs(m.w0) + 1e-5
    np.testing.assert_allclose(w, m.w, rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(v, m.v, rtol=RTOL, atol=1e-5)
    # predictions and the evaluation over the shards
    np.testing.assert_allclose(grp.predict(1, te.n_rows), oracle.predict_raw(m, te), rtol=RTOL, atol=5e-5)
    ev = grp.evaluate(0)
    ref, mae = oracle.evaluate(m, tr, g.task, g.min_target, g.max_target)
    if g.task == 0:
        assert abs(ev.rmse - ref) <= 1e-4 * ref + 1e-6 and abs(ev.mae - mae) <= 1e-4 * mae + 1e-6
    else:
        assert abs(ev.accuracy - ref) * tr.n_rows <= 3
    grp.close()
    for h in hs:
        h.close()


@pytest.mark.parametrize("k", [64, 128, 70])
@pytest.mark.parametrize("world,lag,flags_lag", [(4, 1, True), (8, 2, True), (4, 0, False), (8, 3, True), (3, 1, True)])
@pytest.mark.parametrize("task", [0, 1])
def test_short_rows_of_a_shard_several_examples_per_wavefront(capi, oracle, k, world, lag, flags_lag, task):
    """round 5: a shard of P GPUs sees nnz / P entries per example; at one row per wave-wide load (k >= 64) its sums and its update then run
    several examples per wavefront (k_rowsums_multi, k_apply_multi: S_e staged once per example, rest_e and the multiplier computed in the
    update kernel under the bias-lag schedule) + k_apply_seg over the batch's deferred list.  Ragged rows incl. empty ones and some longer than
    one 32-slot round, ids that repeat inside a batch and inside a row, a ragged last batch, padded factors (k = 70), exact chunk coupling
    (no lag: the multipliers come out of the recurrence) -- the oracle's rule at 1e-4, as every other form of the step."""
    rng = np.random.default_rng(100 * k + world + task)
    n, rows, batch, chunk = 6000, 1900, 512, 32
    sizes = rng.integers(0, 40, rows)
    sizes[rng.random(rows) < 0.05] = 0
    sizes[:3] = (150, 70, 33)                                         # (every shard sees rows beyond one round / beyond the mask's 64 entries)
    rp = np.zeros(rows + 1, dtype=np.uint64)
    rp[1:] = np.cumsum(sizes)
    ent = np.zeros(int(rp[-1]), dtype=capi.ENTRY_DTYPE)
    ent["id"] = (rng.zipf(1.3, len(ent)) % n).astype(np.uint32)        # a head of frequent ids: many deferred features per batch
    far = rng.random(len(ent)) < 0.6
    ent["id"][far] = rng.integers(0, n, int(far.sum()))
    ent["value"] = rng.choice([1.0, 0.5, -1.0, 2.0], len(ent)).astype(np.float32)
    y = (np.where(rng.random(rows) < 0.6, 1.0, -1.0) if task == 1 else rng.normal(0.2, 0.6, rows)).astype(np.float32)
    lo, hi = (-1.0, 1.0) if task == 1 else (float(np.quantile(y, 0.05)), float(np.quantile(y, 0.95)))
    lr, reg = 0.001, (0.0, 0.001, 0.002)                              # (rows of 150 entries with values up to 2: at lr 0.002 the regression runs away)
    d = oracle.Data(ent, rp, y)
    m = oracle.Model(n, k, True, True, *reg)
    m.v[:] = oracle.init_values(3, n, k, 0.01)
    m.w[:] = oracle.init_values(4, n, 1, 0.01)[0]
    m.w0 = 0.02
    hs, grp = make_group(capi, world, [0] * world, n, k, task, reg, lr, lo, hi, 1)
    grp.set_params(m.w0, m.w, m.v)
    grp.upload_rows(0, ent, rp, y)
    flags = capi.FLAG_BIAS_LAG if flags_lag else 0
    for _ in range(2):
        grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, batch, chunk, flags, lag)
        oracle.sgd_epoch_minibatch(m, d, task, lr, lo, hi, batch, chunk, bias_lag=lag if flags_lag else 0)
    w0, w, v = grp.get_params()
    assert abs(w0 - m.w0) <= RTOL * abs(m.w0) + 1e-5
    np.testing.assert_allclose(w, m.w, rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(v, m.v, rtol=RTOL, atol=1e-5)
    assert np.abs(oracle.predict_raw(m, d)).max() < 10.0                # (a trained, bounded model: the comparison below means something)
    np.testing.assert_allclose(grp.predict(0, rows), oracle.predict_raw(m, d), rtol=RTOL, atol=1e-4)
    grp.close()
    for h in hs:
        h.close()


@pytest.mark.parametrize("schedule", ["general", "in_stream", "threads"])
def test_small_batch_schedules_of_the_group_driver_are_one_rule(capi, oracle, monkeypatch, schedule):
    """a 512-row batch over 8 shards (BASELINE configs[2]'s shape of the step) through the three schedules fmx_group_sgd_epoch has for it:
    the general one (~20 host calls per shard and batch: comm-stream events, the recurrence on the side stream), the in-stream one (default:
    3 launches per shard and batch) and the in-stream one with a host thread per shard (FMX_GROUP_THREADS=1) -- the oracle's rule at 1e-4."""
    monkeypatch.setenv("FMX_GROUP_IN_STREAM", "0" if schedule == "general" else "1")
    monkeypatch.setenv("FMX_GROUP_THREADS", "1" if schedule == "threads" else "0")
    import datagen as DG
    rows, k, world, lag = 4000, 64, 8, 2
    e, rp, y, n = DG.criteo_shaped(rows, 21, cat_ids=2000)
    d = oracle.Data(e, rp, y)
    m = oracle.Model(n, k, True, True, 0.0, 0.0005, 0.001)
    m.v[:] = oracle.init_values(3, n, k, 0.05)
    m.w0 = 0.03
    hs, grp = make_group(capi, world, [0] * world, n, k, 1, (0.0, 0.0005, 0.001), 0.01, -1.0, 1.0, 1)
    grp.set_params(m.w0, m.w, m.v)
    grp.upload_rows(0, e, rp, y)
    for _ in range(2):
        st = grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, 0, 0, capi.FLAG_BIAS_LAG, lag)
        assert 64 <= st.batch_used <= 4096 and st.batches >= 4
        oracle.sgd_epoch_minibatch(m, d, 1, 0.01, -1.0, 1.0, st.batch_used, st.w0_chunk_used, bias_lag=lag)
    w0, w, v = grp.get_params()
    assert abs(w0 - m.w0) <= RTOL * abs(m.w0) + 1e-5
    np.testing.assert_allclose(w, m.w, rtol=RTOL, atol=2e-5)
    np.testing.assert_allclose(v, m.v, rtol=RTOL, atol=2e-5)
    grp.close()
    for h in hs:
        h.close()


def test_hashed_ownership_balances_structured_ids(capi):
    """ids that are all multiples of 8 (or all inside one residue class of any small modulus) land on ONE shard under
    `j mod P`; the permutation spreads them"""
    n, world = 1 << 20, 8
    ids = (np.arange(20000, dtype=np.uint32) * 8) % n
    ent = np.zeros(len(ids), dtype=capi.ENTRY_DTYPE)
    ent["id"], ent["value"] = ids, 1.0
    rp = np.arange(0, len(ids) + 1, 4, dtype=np.uint64)
    y = np.zeros(len(rp) - 1, dtype=np.float32)
    counts = {0: [], 1: []}
    for hashed in (0, 1):
        for r in range(world):
            h = capi.Handle(n, 2, True, True, 0, shard_rank=r, shard_world=world, shard_hash=hashed)
            h.upload_rows(0, ent, rp, y)
            counts[hashed].append(len(h.download_rows(0)[0]))
            h.close()
    assert counts[0][0] == len(ids) and sum(counts[0][1:]) == 0          # mod: everything on shard 0
    assert sum(counts[1]) == len(ids)
    assert max(counts[1]) < 1.15 * len(ids) / world and min(counts[1]) > 0.85 * len(ids) / world


@pytest.mark.parametrize("shard_hash", [0, 1])
def test_sharded_params_round_trip_and_rows(capi, oracle, shard_hash):
    """set_params / get_params / get_param_rows / init_params agree between a sharded and an unsharded handle"""
    n, k, world = 5003, 8, 3
    rng = np.random.default_rng(5)
    w, v = rng.normal(0, 1, n), rng.normal(0, 1, (k, n))
    hs = [capi.Handle(n, k, shard_rank=r, shard_world=world, shard_hash=shard_hash) for r in range(world)]
    for h in hs:
        h.set_params(0.5, w, v)
    wo, vo = np.zeros(n), np.zeros((k, n))
    seen = np.zeros(n, dtype=int)
    for h in hs:
        w1, v1 = np.full(n, np.nan), np.full((k, n), np.nan)
        _, w1, v1 = h.get_params(w1, v1)
        own = ~np.isnan(w1)
        seen += own
        assert own.sum() == h.n_local
        wo[own], vo[:, own] = w1[own], v1[:, own]
        ids = np.flatnonzero(own)[:50].astype(np.uint32)
        wr, vr = h.get_param_rows(ids)
        assert np.array_equal(wr, w1[ids]) and np.array_equal(vr, v1[:, ids])
    assert (seen == 1).all()                                              # every feature has exactly one owner
    assert np.array_equal(wo, w.astype(np.float32).astype(np.float64))
    assert np.array_equal(vo, v.astype(np.float32).astype(np.float64))
    # device-side fill: the value of a feature does not depend on how the table is sharded
    full = capi.Handle(n, k)
    full.init_params(0.0, 0.1, 9)
    _, wf, vf = full.get_params()
    for h in hs:
        h.init_params(0.0, 0.1, 9)
    for h in hs:
        w1, v1 = np.full(n, np.nan), np.full((k, n), np.nan)
        _, w1, v1 = h.get_params(w1, v1)
        own = ~np.isnan(w1)
        assert np.array_equal(v1[:, own], vf[:, own])
    full.close()
    for h in hs:
        h.close()


@pytest.mark.parametrize("algo", [0, 1])              # fmx_config::exchange_algo: ncclAllReduce | ncclReduceScatter + ncclAllGather
@pytest.mark.parametrize("world,lag,pipeline", [(1, 2, False), (1, 1, True), (2, 2, False), (4, 1, False)])
def test_library_rccl_schedule_matches_the_oracle_rule(capi, oracle, world, lag, pipeline, algo):
    """the schedule a libFM process runs on several GPUs (fmx_comm_init_rank / a group of handles on distinct devices ->
    fmx_sgd_epoch): partial sums and their all-reduce in RUNS of rows (the wire works while the next run is summed), then the
    update.  world = 1: one rank through the RCCL binding (the all-reduce is the identity, every offset of the chunked
    exchange is still exercised: batches of 1000 rows = runs of 256 / 256 / 256 / 232, ragged last batch); world > 1: one
    handle per GPU from this one process, where that many GPUs are visible."""
    import torch
    if world > torch.cuda.device_count():
        pytest.skip("needs %d GPUs" % world)
    n, k, nnz, rows, batch, chunk = 5000, 16, 12, 3300, 1000, 50
    ent, rp, y = datagen.onehot_fields(n, nnz, rows, seed=77, zipf=1.1)
    m = oracle.Model(n, k, True, True, 0.0, 0.002, 0.01)
    m.v[:] = oracle.init_values(8, n, k, 0.1)
    m.w[:] = oracle.init_values(9, n, 1, 0.1)[0]
    m.w0 = 0.05
    d = oracle.Data(ent, rp, y)
    hs = [capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.002, 0.01, 0.004, -1.0, 1.0, device=r, shard_rank=r,
                      shard_world=world, shard_hash=1, exchange_algo=algo) for r in range(world)]
    for h in hs:
        h.set_params(m.w0, m.w, m.v)
        h.upload_rows(0, ent, rp, y)
    flags = capi.FLAG_BIAS_LAG | (capi.FLAG_PIPELINE if pipeline else 0)
    if world == 1:
        hs[0].comm_init_rank(capi.comm_unique_id(), 0, 1)
        run = lambda: hs[0].sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, batch, chunk, flags, lag)
        grp = None
    else:
        grp = capi.Group(hs)
        run = lambda: grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, batch, chunk, flags, lag)
    for _ in range(3):
        run()
        oracle.sgd_epoch_minibatch(m, d, 1, 0.004, -1.0, 1.0, batch, chunk, bias_lag=lag, pipelined=pipeline)
    w0, w, v = grp.get_params() if grp else hs[0].get_params()
    assert np.abs(m.v).max() < 10.0 and np.abs(m.v - oracle.init_values(8, n, k, 0.1)).max() > 1e-3      # trained, not diverged
    assert abs(w0 - m.w0) <= RTOL * abs(m.w0) + 1e-5
    np.testing.assert_allclose(w, m.w, rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(v, m.v, rtol=RTOL, atol=1e-5)
    if grp:
        grp.close()
    for h in hs:
        h.close()


def test_group_set_params_makes_the_weight_side_stream_stale(capi, oracle):
    """round-4 advisor: a ONE-handle group forwards FMX_FLAG_KEEP_WSIDE and predict to its member; fmx_group_set_params wrote new linear weights
    without invalidating the slot's weight side stream, so the next pass read the OLD w_j for every last-occurrence entry.  Epoch that keeps
    the stream -> new parameters through the group -> predictions are the new model's (the oracle's)."""
    n, k, nnz, rows = 3000, 8, 6, 1200
    ent, rp, y = datagen.onehot_fields(n, nnz, rows, seed=5, classification=True)
    d = oracle.Data(ent, rp, y)
    m = oracle.Model(n, k, True, True, 0.0, 0.001, 0.002)
    m.v[:] = oracle.init_values(9, n, k, 0.05)
    m.w[:] = oracle.init_values(10, n, 1, 0.05)[0]
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.001, 0.002, 0.01, -1.0, 1.0)
    grp = capi.Group([h])
    grp.set_params(m.w0, m.w, m.v)
    grp.upload_rows(0, ent, rp, y)
    grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 256, 32, capi.FLAG_KEEP_WSIDE, 2)
    assert h.evaluate(0).flags & capi.EVAL_WSIDE                      # the stream is in use ...
    m2 = oracle.Model(n, k, True, True, 0.0, 0.001, 0.002)
    m2.v[:] = oracle.init_values(11, n, k, 0.05)
    m2.w[:] = oracle.init_values(12, n, 1, 0.5)[0]                     # (large linear weights: a stale stream would be far off)
    m2.w0 = -0.3
    grp.set_params(m2.w0, m2.w, m2.v)
    assert not (h.evaluate(0).flags & capi.EVAL_WSIDE)                # ... and stale after the group wrote new weights
    np.testing.assert_allclose(grp.predict(0, rows), oracle.predict_raw(m2, d), rtol=1e-4, atol=2e-5)
    grp.close()
    h.close()


def test_reduce_scatter_all_gather_is_the_same_sum_on_loopback_shards(capi):
    """fmx_config::exchange_algo = FMX_EXCHANGE_RS_AG on shards that share a device: the two-phase reduction (every shard reduces its slice,
    then everybody copies the slices) leaves the same floats as the one-kernel all-reduce -- the trained models are bit-identical"""
    n, k, nnz, rows, batch, world = 200_000, 64, 16, 40_000, 8192, 4
    res = []
    for algo in (0, 1):
        hs = [capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0, shard_rank=r, shard_world=world,
                          shard_hash=1, exchange_algo=algo) for r in range(world)]
        for x in hs:
            x.init_params(0.0, 0.05, 3)
            x.synth_rows(0, 11, 0, rows, nnz)
        grp = capi.Group(hs)
        for _ in range(2):
            grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, batch, 0, capi.FLAG_BIAS_LAG, 2)
        res.append((grp.predict(0, rows), hs[0].get_w0()))
        grp.close()
        for x in hs:
            x.close()
    assert res[0][0].tobytes() == res[1][0].tobytes() and res[0][1] == res[1][1]
    assert np.abs(res[0][0]).max() > 1e-3


def test_group_at_bench_shape_matches_single_handle(capi):
    """n = 1e7, k = 64, 32 nnz: 4 loopback shards == one handle running the one-pass form of the same rule"""
    n, k, nnz, rows, batch = 10_000_000, 64, 32, 1 << 17, 32768
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 3)
    h.synth_rows(0, 11, 0, rows, nnz)
    h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, batch, 0, 0, 2)
    p_one = h.predict(0, rows)
    w0_one = h.get_w0()
    h.close()
    world = 4
    hs = [capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0, shard_rank=r,
                      shard_world=world, shard_hash=1) for r in range(world)]
    for x in hs:
        x.init_params(0.0, 0.05, 3)
        x.synth_rows(0, 11, 0, rows, nnz)
    grp = capi.Group(hs)
    grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, batch, 0, 0, 2)
    p_grp = grp.predict(0, rows)
    assert abs(hs[0].get_w0() - w0_one) <= 1e-5 * abs(w0_one) + 1e-7
    np.testing.assert_allclose(p_grp, p_one, rtol=1e-4, atol=2e-5)
    grp.close()
    for x in hs:
        x.close()


# ---------------------------------------------------------------------------------------------
# ALS / MCMC over feature shards (fmx_group_als_*): global dependency levels, replicated {e, q} cache, one all-reduce per
# (coordinate family, level).  Bar: the REAL reference's ALS results (golden fixtures) at 1e-4, whatever the shard count;
# MCMC: the sharded chain draws what the unsharded chain draws (noise keyed by global feature id).
# ---------------------------------------------------------------------------------------------
ALS_CASES = ["als_reg_ml", "als_cls_ragged", "als_reg_fields_k16", "als_reg_nolin_dup", "als_reg_ml_groups"]


def _als_group_run(capi, g, oracle, world, shard_hash, do_sample=False, seed=0, iters=None):
    z = g.z
    m = g.model(oracle, "init")
    tr, te = g.data(oracle, "train"), g.data(oracle, "test")
    hs = [capi.Handle(g.n, g.k, g.k0, g.k1, g.task, g.reg[0], g.reg[1], g.reg[2], 0.0, g.min_target, g.max_target, device=0,
                      shard_rank=r, shard_world=world, shard_hash=shard_hash) for r in range(world)]
    for h in hs:
        h.set_params(m.w0, m.w, m.v)
        if "group" in z.files:
            h.set_groups(z["group"])
        h.upload_rows(0, tr.entries, tr.row_ptr, tr.target)
        h.upload_rows(1, te.entries, te.row_ptr, te.target)
    grp = capi.Group(hs) if world > 1 else None
    drv = grp if grp else hs[0]
    wl, vl = (z["w_lambda_g"], z["v_lambda_g"][:, None] * np.ones((1, max(g.k, 1)))) if "group" in z.files else (g.reg[1], g.reg[2])
    drv.als_begin(0)
    metrics = []
    for i in range(iters or g.iters):
        st = drv.als_sweep(wl, vl, do_sample=do_sample, seed=seed)
        metrics.append(st.train_metric)
    pred = grp.predict(1, te.n_rows) if grp else hs[0].predict(1, te.n_rows)
    mom = drv.als_moments()
    drv.als_end()
    w0, w, v = grp.get_params() if grp else hs[0].get_params()
    if grp:
        grp.close()
    for h in hs:
        h.close()
    return w0, w, v, pred, metrics, mom


@pytest.mark.parametrize("world,shard_hash", [(2, 0), (3, 1), (8, 1)])
@pytest.mark.parametrize("name", ALS_CASES)
def test_sharded_als_matches_reference(capi, oracle, name, world, shard_hash):
    g = Golden(name)
    z = g.z
    w0, w, v, pred, metrics, _ = _als_group_run(capi, g, oracle, world, shard_hash)
    assert abs(w0 - float(z["final_w0"])) <= 1e-4 * abs(float(z["final_w0"])) + 2e-5
    np.testing.assert_allclose(w, z["final_w"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v, z["final_v"], rtol=1e-4, atol=2e-5)
    if g.task == 0:                                            # y-hat of the last iteration, clamped like fm_learn_mcmc::predict (:380-400)
        np.testing.assert_allclose(np.clip(pred, g.min_target, g.max_target), z["pred_out"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("name", ["mcmc_reg_ml", "mcmc_cls_fields"])
def test_sharded_mcmc_draws_what_the_unsharded_chain_draws(capi, oracle, name):
    """do_sample = 1 with fixed hyper-parameters: 4 hashed shards vs one handle, same seed -> same chain (the Gibbs noise of a
    coordinate is a hash of (seed, iteration, family, GLOBAL feature id); the probit targets a hash of the row)."""
    g = Golden(name)
    one = _als_group_run(capi, g, oracle, 1, 0, do_sample=True, seed=77, iters=6)
    four = _als_group_run(capi, g, oracle, 4, 1, do_sample=True, seed=77, iters=6)
    assert abs(one[0] - four[0]) <= 1e-6 * abs(one[0]) + 1e-7
    np.testing.assert_allclose(four[1], one[1], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(four[2], one[2], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(four[3], one[3], rtol=1e-4, atol=2e-4)   # fp32 factor sums: 4 partial sums vs one (|v| reaches ~10 in a chain)
    np.testing.assert_allclose(four[4], one[4], rtol=1e-6)
    # the statistics of the hyper-prior draws: residual sums from the replicated cache, parameter sums added over the shards
    np.testing.assert_allclose(four[5][0], one[5][0], rtol=1e-6)
    np.testing.assert_allclose(four[5][2], one[5][2], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("world,shard_hash", [(2, 1), (3, 0), (5, 1)])
def test_group_staging_equals_per_shard_upload(capi, oracle, world, shard_hash):
    """fmx_group_set_params / fmx_group_upload_rows (the host block and the rows cross PCIe once; every shard converts / filters its
    own features on its device) leave every shard with exactly what the per-handle calls with the full arrays leave: same local
    rows (ragged rows, empty rows, repeated ids), same parameters; ids beyond num_attribute are refused."""
    ent, rp, y = datagen.ragged_real(500, 700, 9, seed=3, empty_every=6, duplicates=True)
    n, k = 500, 12
    m = oracle.Model(n, k, True, True, 0.0, 0.0, 0.001)
    m.w0 = 0.25
    m.w[:] = oracle.init_values(7, n, 1, 0.3)[0]
    m.v[:] = oracle.init_values(8, n, k, 0.1)
    a, ga = make_group(capi, world, [0] * world, n, k, capi.TASK_CLASSIFICATION, (0.0, 0.0, 0.001), 0.01, -1.0, 1.0, shard_hash)
    b, gb = make_group(capi, world, [0] * world, n, k, capi.TASK_CLASSIFICATION, (0.0, 0.0, 0.001), 0.01, -1.0, 1.0, shard_hash)
    for h in a:
        h.set_params(m.w0, m.w, m.v)
        h.upload_rows(0, ent, rp, y)
    gb.set_params(m.w0, m.w, m.v)
    gb.upload_rows(0, ent, rp, y)
    for ha, hb in zip(a, b):
        ea, ra, ya = ha.download_rows(0)
        eb, rb, yb = hb.download_rows(0)
        assert np.array_equal(ea, eb) and np.array_equal(ra, rb) and np.array_equal(ya, yb)
    wa, wb = ga.get_params(), gb.get_params()
    assert wa[0] == wb[0] and np.array_equal(wa[1], wb[1]) and np.array_equal(wa[2], wb[2])
    np.testing.assert_allclose(wb[2], m.v, rtol=1e-6, atol=1e-7)
    # and they train alike
    sa = ga.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, 64, 16, capi.FLAG_BIAS_LAG, 1)
    sb = gb.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, 64, 16, capi.FLAG_BIAS_LAG, 1)
    pa, pb = ga.get_params(), gb.get_params()
    assert np.array_equal(pa[2], pb[2]) and np.array_equal(pa[1], pb[1]) and sa.max_feature_count == sb.max_feature_count
    bad = ent.copy()
    bad["id"][5] = n + 3
    with pytest.raises(capi.FmxError):
        gb.upload_rows(1, bad, rp, y)
    for g_ in (ga, gb):
        g_.close()
    for h in a + b:
        h.close()
