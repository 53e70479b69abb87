"""GPU: every BASELINE.json config at its OWN shape against the oracle (round-3 verdict, "missing 1").

The models of configs[1..4] do not fit a host-side oracle run (n = 1e7 ... 1e8), so parity is held on the SUB-MODEL of the rows'
features -- the method of tests/test_gpu_fullsize.py: the parameter rows the rows touch are fetched from the device BEFORE the
step (fmx_get_param_rows), the ids are remapped to a dense small model (order-preserving, so the coordinate order of an ALS sweep
is kept), the oracle runs the same step on it, and the device's rows after the step are compared at 1e-4.  Untouched parameters do
not enter the arithmetic of any of these learners.

  configs[1]  n = 1e7,  k = 32, 16 entries/row, SGD (one-pass batch rule, the library's batch, bias lag 2)
  configs[2]  n = 3.3e7, k = 64, 39 entries/row, Criteo-shaped ids (fmx_synth_rows_ex(FMX_SYNTH_CRITEO)): the library cuts the batch
  configs[3]  ALS (fm_learn_mcmc, do_sample = 0) at k = 64, n = 1e7
  configs[4]  the same sweep at k = 128, n = 1e8 (the chain's deterministic part; the sampled chain is held statistically,
              tests/test_gpu_mcmc.py)
and the headline rule's distance from the reference's ONLINE loop (fm_learn_sgd_element.h:56-67) at the bench shape, as a band."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import capi as c
    if c.load().fmx_device_count() == 0:
        pytest.fail("no HIP device: the GPU tests must run on the MI355X box")
    return c


def sub_model(oracle, h, ent, rp, y, k, reg):
    """dense copy of the rows and of the parameter rows they touch (as the device holds them now)"""
    ids = np.unique(ent["id"])
    w, v = h.get_param_rows(ids)
    e2 = ent.copy()
    e2["id"] = np.searchsorted(ids, ent["id"]).astype(np.uint32)
    m = oracle.Model(len(ids), k, True, True, *reg)
    m.w[:], m.v[:], m.w0 = w, v, h.get_w0()
    return oracle.Data(e2, rp, y), m, ids


def assert_rows(h, m, ids, atol=1e-6):
    w, v = h.get_param_rows(ids)
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=atol)
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=atol)
    assert abs(h.get_w0() - m.w0) <= 1e-4 * abs(m.w0) + atol


def test_config1_sgd_n1e7_k32_nnz16(capi, oracle):
    """BASELINE configs[1]: 'Synthetic 1e7 features, k=32, nnz=16/row, SGD on 1xMI355X' -- k_fused<32, 8, EXACT> (two rows per
    wave-wide load) + the deferred features of a 262 144-row batch (2.5 per example at this id space), bias lag 2, library batch"""
    n, k, nnz, rows, lag = 10_000_000, 32, 16, 262144 + 8192, 2
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 17)
    h.synth_rows(0, 606, 3_000_000, rows, nnz)
    src = oracle.synth_rows(606, 3_000_000, rows, nnz, n)
    d, m, ids = sub_model(oracle, h, src.entries, src.row_ptr, src.target, k, (0.0, 0.0, 0.001))
    st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, lag)
    assert st.batch_used == 262144 and st.status & capi.STAT_WARN == 0 and st.batches == 2
    assert st.deferred_features > 1.5 * rows                      # most examples leave their sums behind at this shape
    oracle.sgd_epoch_minibatch(m, d, 1, 0.01, -1.0, 1.0, 262144, st.w0_chunk_used, bias_lag=lag)
    assert_rows(h, m, ids)
    np.testing.assert_allclose(h.predict(0, rows), oracle.predict_raw(m, d), rtol=1e-4, atol=2e-5)
    h.close()


def test_config2_sgd_criteo_shaped_at_size(capi, oracle):
    """BASELINE configs[2] on one GPU: n = 3.3e7, k = 64, 39 entries/row, 13 fields of <= 100 ids + 26 Zipf(1.05) fields, batch 0:
    the library cuts the batch to the rows' stability bound (512 at lr 0.01) and runs >= 64 of them through the small-batch path
    (k_fused<64, 40, EXACT> + deferred features and recurrence in the stream) -- against the oracle's rule at that batch"""
    n, k, nnz, rows, lag = 33_000_000, 64, 39, 40_000, 2
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 19)
    h.synth_rows(0, 123, 500_000, rows, nnz, capi.SYNTH_CRITEO)
    ent, rp, y = h.download_rows(0)
    assert len(y) == rows and len(ent) == rows * nnz
    d, m, ids = sub_model(oracle, h, ent, rp, y, k, (0.0, 0.0, 0.001))
    bi = h.sgd_batch_info(0)
    assert bi.status & capi.STAT_BATCH_CUT and 128 <= bi.batch <= 1024 and bi.batch_gain <= 1.0
    for _ in range(2):                                            # second epoch: every frequent feature has moved
        st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, lag)
        assert st.batch_used == bi.batch and st.batches >= 64 and st.deferred_features > rows
        oracle.sgd_epoch_minibatch(m, d, 1, 0.01, -1.0, 1.0, bi.batch, st.w0_chunk_used, bias_lag=lag)
    assert_rows(h, m, ids, atol=2e-5)
    np.testing.assert_allclose(h.predict(0, rows), oracle.predict_raw(m, d), rtol=1e-4, atol=5e-5)
    h.close()


def test_reference_trajectory_on_criteo_shaped_rows_at_size(capi, oracle):
    """FMX_SGD_SEQUENTIAL -- the reference's own loop (fm_learn_sgd_element.h:56-67: one example at a time, file order) -- at BASELINE
    configs[2]'s size: every example shares its 13 dense-field features with its neighbours, so k_sequential_wg reads those rows a
    second time behind its second barrier for nearly every example; two epochs == the oracle's ONLINE loop on the sub-model at 1e-4."""
    n, k, nnz, rows = 33_000_000, 64, 39, 12_000
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 23)
    h.synth_rows(0, 77, 250_000, rows, nnz, capi.SYNTH_CRITEO)
    ent, rp, y = h.download_rows(0)
    d, m, ids = sub_model(oracle, h, ent, rp, y, k, (0.0, 0.0, 0.001))
    for _ in range(2):
        h.sgd_epoch(0, capi.SGD_SEQUENTIAL)
        oracle.sgd_epoch_online(m, d, 1, 0.01, -1.0, 1.0)
    assert_rows(h, m, ids, atol=2e-5)
    np.testing.assert_allclose(h.predict(0, rows), oracle.predict_raw(m, d), rtol=1e-4, atol=5e-5)
    h.close()


def test_config2_eight_shards_criteo_shaped(capi, oracle):
    """BASELINE configs[2] AS IT IS WORDED: 'Criteo-Kaggle-shaped: 3.3e7 features, k = 64, 39 entries per row, SGD, V row-sharded across 8':
    eight feature shards (on one GPU: the loopback exchange) driven by fmx_group_sgd_epoch at the library's batch (512 rows at lr 0.01: the
    in-stream schedule of small batches -- three launches per shard and batch) against ONE unsharded handle on the same rows and start
    values, and against the oracle's rule on the sub-model of the touched features (1e-4).  The loop it stands for:
    fm_learn_sgd_element.h:56-67."""
    n, k, nnz, rows, lag, world = 33_000_000, 64, 39, 20_000, 2, 8
    one = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    one.init_params(0.0, 0.05, 19)
    one.synth_rows(0, 123, 500_000, rows, nnz, capi.SYNTH_CRITEO)
    ent, rp, y = one.download_rows(0)
    d, m, ids = sub_model(oracle, one, ent, rp, y, k, (0.0, 0.0, 0.001))
    bi = one.sgd_batch_info(0)
    hs = [capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0, shard_rank=r, shard_world=world, shard_hash=1)
          for r in range(world)]
    for h in hs:
        h.init_params(0.0, 0.05, 19)                              # (keyed by the GLOBAL feature id: a shard draws what the unsharded handle draws)
        h.synth_rows(0, 123, 500_000, rows, nnz, capi.SYNTH_CRITEO)
    grp = capi.Group(hs)
    np.testing.assert_allclose(grp.predict(0, rows), one.predict(0, rows), rtol=1e-4, atol=2e-5)     # same start
    for _ in range(2):
        st1 = one.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, lag)
        st8 = grp.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, 0, 0, capi.FLAG_BIAS_LAG, lag)
        assert st8.batch_used == st1.batch_used == bi.batch and 128 <= bi.batch <= 1024 and st8.batches == st1.batches >= 32
        oracle.sgd_epoch_minibatch(m, d, 1, 0.01, -1.0, 1.0, bi.batch, st1.w0_chunk_used, bias_lag=lag)
    p1, p8, po = one.predict(0, rows), grp.predict(0, rows), oracle.predict_raw(m, d)
    assert np.abs(po).max() > 0.05
    np.testing.assert_allclose(p1, po, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(p8, po, rtol=1e-4, atol=5e-5)
    assert abs(hs[3].get_w0() - m.w0) <= 1e-4 * abs(m.w0) + 2e-5
    grp.close()
    for h in hs:
        h.close()
    one.close()


def als_sub_model_sweeps(capi, oracle, n, k, nnz, rows, seed, sweeps, stdev=0.05):
    """`sweeps` iterations of fm_learn_mcmc with do_sample = 0 (fmx_als_sweep; split step on: every level holds `rows` >= 65 536
    entries) on the full-size table vs the oracle's learner (fmo_als_learn: fm_learn_mcmc.h:430-641 + _learn) on the sub-model"""
    w_lambda, v_lambda = 1.0, 10.0
    h = capi.Handle(n, k, True, True, capi.TASK_REGRESSION, 0.0, w_lambda, v_lambda, 0.0, -1.0, 1.0)
    h.init_params(0.0, stdev, seed)
    h.synth_rows(0, 900 + seed, 1_000_000, rows, nnz)
    h.synth_rows(1, 900 + seed, 5_000_000, 2048, nnz)             # test rows: other features, mostly without a training column
    src = oracle.synth_rows(900 + seed, 1_000_000, rows, nnz, n)
    d, m, ids = sub_model(oracle, h, src.entries, src.row_ptr, src.target, k, (0.0, w_lambda, v_lambda))
    h.als_begin(0)
    metric_dev = [h.als_sweep(w_lambda, v_lambda).train_metric for _ in range(sweeps)]
    p_train = h.predict(0, rows)
    h.als_end()
    test = oracle.Data(d.entries[:nnz * 64], d.row_ptr[:65], d.target[:64])
    _, metric = oracle.als_learn(m, d, test, 0, sweeps, w_lambda, v_lambda, -1.0, 1.0)
    w, v = h.get_param_rows(ids)
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=2e-5)
    assert abs(h.get_w0() - m.w0) <= 1e-4 * abs(m.w0) + 2e-5
    np.testing.assert_allclose(metric_dev, metric, rtol=1e-4)
    np.testing.assert_allclose(p_train, oracle.predict_raw(m, d), rtol=1e-4, atol=5e-5)
    h.close()


def test_config3_als_k64_n1e7(capi, oracle):
    """BASELINE configs[3]: 'ALS learner (fm_learn_als): k=64' at the bench's n = 1e7, 16 entries/row: two sweeps over 131 072 rows
    (16 levels x 65 coordinate families, row-ordered update) == the oracle's sweep on the 2.1 M touched features"""
    als_sub_model_sweeps(capi, oracle, 10_000_000, 64, 16, 131072, 5, 2)


def test_config4_sweep_k128_n1e8(capi, oracle):
    """BASELINE configs[4]'s shape: k = 128 (two floats per lane), n = 1e8 (a 51 GB table), 16 entries/row -- the sweep without
    sampling (do_sample = 0), which is the chain's arithmetic minus the noise: one sweep over 98 304 rows == the oracle"""
    als_sub_model_sweeps(capi, oracle, 100_000_000, 128, 16, 98304, 7, 1)


def test_config4_eight_shards_k128_n1e8_equal_one_handle(capi):
    """BASELINE configs[4] as it is worded: 'k = 128, 1e8 features, V sharded across 8': the sweep through fmx_group_als_* over EIGHT feature
    shards (on one GPU: the loopback exchange; 8 x 6.4 GB of tables next to the single handle's 51 GB) against the single-handle sweep
    on the same rows and start values -- without sampling and WITH (do_sample: the draws are keyed by the GLOBAL feature id, so a
    shard draws what the unsharded handle draws).  The shards re-predict from all-reduced fp64 caches; agreement is to fp32 rounding."""
    n, k, nnz, rows, world = 100_000_000, 128, 16, 65536, 8
    w_lambda, v_lambda = 1.0, 10.0
    res = {}
    for tag in ("one", "eight"):
        W = 1 if tag == "one" else world
        hs = [capi.Handle(n, k, True, True, capi.TASK_REGRESSION, 0.0, w_lambda, v_lambda, 0.0, -1.0, 1.0, device=0, shard_rank=r, shard_world=W,
                          shard_hash=1 if W > 1 else 0) for r in range(W)]
        for h in hs:
            h.init_params(0.0, 0.05, 7)
            h.synth_rows(0, 4242, 2_000_000, rows, nnz)
        out = []
        if W == 1:
            h = hs[0]
            h.als_begin(0)
            out.append(h.als_sweep(w_lambda, v_lambda).train_metric)
            out.append(h.als_sweep(w_lambda, v_lambda, alpha=1.5, do_sample=True, seed=99).train_metric)
            p = h.predict(0, rows)
            h.als_end()
        else:
            grp = capi.Group(hs)
            grp.als_begin(0)
            out.append(grp.als_sweep(w_lambda, v_lambda).train_metric)
            out.append(grp.als_sweep(w_lambda, v_lambda, alpha=1.5, do_sample=True, seed=99).train_metric)
            p = grp.predict(0, rows)
            grp.als_end()
            grp.close()
        res[tag] = (np.array(out), p)
        for h in hs:
            h.close()
    np.testing.assert_allclose(res["eight"][0], res["one"][0], rtol=1e-4)
    np.testing.assert_allclose(res["eight"][1], res["one"][1], rtol=1e-4, atol=5e-5)
    assert np.abs(res["one"][1]).max() > 1e-3


# the band DESIGN.md section 3 states (measured on the CPU with the oracle's two loops: scripts/cpu_online_vs_rule.py, and asserted
# here with the DEVICE in place of the oracle's rule)
# measured over these 278 528 rows (1.06 batches): micro-chunk 256 (round 4, profiles/r04_parity_vs_online.json) bias 0.0247 apart, predictions (rms 0.20)
# mean 0.024 / max 0.049; micro-chunk 32 (the default since round 5; the round-4 verdict's run of scripts/cpu_online_vs_rule.py) bias 0.0014, predictions
# mean 0.0056 / max 0.027, factors and weights as before (max 1.6e-4 = 0.8 % of max |v|, 1.1e-3).  The band is that x 1.6-2.
ONLINE_BAND = {"w0_abs": 0.0028, "pred_mean_abs": 0.0095, "pred_max_abs": 0.045, "pred_max_rel_to_rms_without_bias": 0.2,
               "v_max_rel_to_vmax": 0.016, "w_max_abs": 0.002}


def test_headline_rule_vs_the_online_loop_at_the_bench_shape(capi, oracle):
    """north_star: 'predictions within 1e-4 relative of the CPU reference'.  The headline mode is held to 1e-4 against the restated
    batch rule (tests/test_gpu_fullsize.py); the reference itself is ONLINE (batch 1, fm_learn_sgd_element.h:56-67).  This test puts
    a number on the distance between the two at the bench configuration -- n = 1e8, k = 64, 32 entries/row, batch 262 144, bias lag
    2, bench.py's start values (stdev 0.01) -- after one epoch over 278 528 rows: the device's rows and predictions against the
    oracle's ONLINE loop on the sub-model, inside the stated band; and against the oracle's rule at 1e-4 (same run)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from cpu_online_vs_rule import deviation
    n, k, nnz, rows, lag = 100_000_000, 64, 32, 262144 + 16384, 2
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    src = oracle.synth_rows(123, 0, rows, nnz, n)
    d, m_rule, ids = sub_model(oracle, h, src.entries, src.row_ptr, src.target, k, (0.0, 0.0, 0.001))
    m_on = m_rule.copy()
    h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, lag)
    oracle.sgd_epoch_online(m_on, d, 1, 0.01, -1.0, 1.0)
    oracle.sgd_epoch_minibatch(m_rule, d, 1, 0.01, -1.0, 1.0, 262144, capi.default_w0_chunk(0.01, 1), bias_lag=lag)
    assert_rows(h, m_rule, ids)                                   # the device IS the rule ...
    m_dev = m_rule.copy()
    m_dev.w[:], m_dev.v[:] = h.get_param_rows(ids)
    m_dev.w0 = h.get_w0()
    dev = deviation(oracle, m_dev, m_on, d)                       # ... and the rule ends this far from the online loop
    print("parity_vs_online", dev)
    for key, bound in ONLINE_BAND.items():
        assert dev[key] <= bound, (key, dev[key], bound)
    assert dev["v_max_abs"] > 0                                   # (the two loops are different iterates)
    h.close()
