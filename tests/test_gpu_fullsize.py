"""GPU, at BASELINE.json's FULL size (n = 1e8 features, k = 64, 32 nnz/row; 25.6 GB of parameters): the fm_model of
that size does not fit a host-side oracle run, so parity is held through
  (1) a spot check against the oracle on a SUB-MODEL: for a sample of rows, the touched parameter rows are fetched
      (fmx_get_param_rows), remapped to a dense small model, and fm_model::predict / one online fm_SGD epoch of the
      oracle are compared with what the device computed for exactly those rows;
  (2) size-independent properties: degree-2 homogeneity of the pairwise term, run-to-run determinism of the
      segmented minibatch step, feature shards summing to the unsharded partial sums."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N, K, NNZ = 100_000_000, 64, 32


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import capi
    return capi


def small_problem(capi, oracle, h, seed, row0, rows):
    """oracle-sized copy of `rows` synthetic rows and of the parameters they touch"""
    d = oracle.synth_rows(seed, row0, rows, NNZ, N)
    ids = np.unique(d.entries["id"])
    w, v = h.get_param_rows(ids)
    remap = np.searchsorted(ids, d.entries["id"]).astype(np.uint32)
    ent = d.entries.copy()
    ent["id"] = remap
    m = oracle.Model(len(ids), K, True, True, 0.0, 0.0, 0.001)
    m.w[:], m.v[:], m.w0 = w, v, h.get_w0()
    return oracle.Data(ent, d.row_ptr, d.target), m, ids


def test_predict_spot_check_against_oracle(capi, oracle):
    h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 7)
    rows = 4096
    h.synth_rows(0, 123, 1_000_000, rows, NNZ)                # rows 1 000 000 .. of the synthetic stream
    p = h.predict(0, rows)
    d, m, _ = small_problem(capi, oracle, h, 123, 1_000_000, rows)
    np.testing.assert_allclose(p, oracle.predict_raw(m, d), rtol=1e-4, atol=2e-5)
    h.close()


def test_training_step_spot_check_against_oracle(capi, oracle):
    """one SEQUENTIAL epoch and one segmented MINIBATCH epoch over 2048 rows of the full-size model == the oracle on
    the sub-model of the touched rows (untouched parameters do not enter the arithmetic)."""
    rows = 2048
    for mode, batch in ((capi.SGD_SEQUENTIAL, 0), (capi.SGD_MINIBATCH, 256)):
        h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
        h.init_params(0.0, 0.05, 7)
        h.synth_rows(0, 321, 5_000_000, rows, NNZ)
        d, m, ids = small_problem(capi, oracle, h, 321, 5_000_000, rows)
        if mode == capi.SGD_SEQUENTIAL:
            h.sgd_epoch(0, mode)
            oracle.sgd_epoch_online(m, d, 1, 0.01, -1.0, 1.0)
        else:
            h.sgd_epoch(0, mode, capi.APPLY_SEGMENTED, batch, 64)
            oracle.sgd_epoch_minibatch(m, d, 1, 0.01, -1.0, 1.0, batch, 64)
        w, v = h.get_param_rows(ids)
        np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=1e-6)
        assert abs(h.get_w0() - m.w0) <= 1e-4 * abs(m.w0) + 1e-6
        h.close()


def submodel_minibatch(capi, oracle, h, seed, row0, rows, batch, chunk, lag):
    """the oracle's batch rule on the sub-model of the rows' features (parameters fetched BEFORE the device step)"""
    d, m, ids = small_problem(capi, oracle, h, seed, row0, rows)
    oracle.sgd_epoch_minibatch(m, d, 1, 0.01, -1.0, 1.0, batch, chunk, bias_lag=lag)
    return d, m, ids


def test_fused_minibatch_spot_check_at_full_size(capi, oracle):
    """the bench mode (MINIBATCH rule, FMX_APPLY_FUSED, bias_lag 2) on the 25.6 GB model: 4 batches of 16 384 rows ==
    the oracle's rule on the sub-model of the touched rows, colliding features included (about 0.5 % of the entries of
    a batch share their feature with another example: those go through the segmented kernel)."""
    rows, batch, chunk, lag = 65536, 16384, 256, 2
    h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 7)
    h.synth_rows(0, 4321, 9_000_000, rows, NNZ)
    d, m, ids = submodel_minibatch(capi, oracle, h, 4321, 9_000_000, rows, batch, chunk, lag)
    st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, batch, chunk, 0, lag)
    assert 0 < st.deferred_features < 0.02 * rows * NNZ
    w, v = h.get_param_rows(ids)
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=1e-6)
    assert abs(h.get_w0() - m.w0) <= 1e-4 * abs(m.w0) + 1e-6
    h.close()


def test_fused_minibatch_at_the_bench_batch_against_oracle(capi, oracle):
    """the configuration BENCH reports -- n = 1e8, batch 262 144, bias lag 2, library-default micro-chunk -- against the oracle's
    rule: one full batch (1.27 deferred features per example, 8.3 M touched parameter rows: a 4.3 GB fp64 sub-model on the host)
    plus a short second one, 1e-4."""
    rows, batch, lag = 262144 + 16384, 262144, 2
    h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 11)
    h.synth_rows(0, 2024, 40_000_000, rows, NNZ)
    d, m, ids = submodel_minibatch(capi, oracle, h, 2024, 40_000_000, rows, batch, capi.default_w0_chunk(0.01, 1), lag)
    st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, lag)       # batch 0 / chunk 0: what bench.py passes
    assert st.batch_used == batch and st.status & capi.STAT_WARN == 0 and st.batches == 2 and st.w0_chunk_used == capi.default_w0_chunk(0.01, 1) <= 64
    assert 1.0 * rows < st.deferred_features < 1.5 * rows
    w, v = h.get_param_rows(ids)
    np.testing.assert_allclose(v, m.v, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w, m.w, rtol=1e-4, atol=1e-6)
    assert abs(h.get_w0() - m.w0) <= 1e-4 * abs(m.w0) + 1e-6
    h.close()


def test_fused_minibatch_is_deterministic_at_bench_batch(capi):
    """two runs of the bench configuration (batch 262 144, bias_lag 2) give bit-identical predictions and bias -- and so does a third
    one that orders the launch stream and the recurrence's side stream with events instead of the device-side hand-off
    (FMX_FLAG_EVENT_SYNC): the hand-off moves no number"""
    rows = 1 << 20
    sums = []
    for flags in (0, 0, capi.FLAG_EVENT_SYNC):
        h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
        h.init_params(0.0, 0.05, 3)
        h.synth_rows(0, 77, 0, rows, NNZ)
        for _ in range(2):                                        # (second epoch: the hand-off counter carries on, the slots are re-armed)
            st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 262144, 0, flags, 2)
        p = h.predict(0, rows)
        sums.append((p.tobytes(), h.get_w0(), st.deferred_features))
        h.close()
    assert sums[0] == sums[1] == sums[2]
    assert 0.02 * rows * NNZ < sums[0][2] < 0.06 * rows * NNZ   # ~4 % of the (batch, feature) pairs hold >= 2 occurrences


def test_hogwild_at_bench_size_is_the_batch_rule_off_collisions(capi, oracle):
    """HOGWILD where it is benchmarked (n = 1e8, one 65 536-row launch, bias frozen for the launch).  An example none of
    whose features occurs anywhere else in the launch is independent of every other example, so its parameter rows must
    come out exactly as the batch rule leaves them (oracle, batch = launch).  Examples that share a feature with another
    one race: a racing reader may see its partner's update, which perturbs that example's sums and with them ALL of its
    rows -- those are counted and bounded, not held to 1e-4."""
    rows = 65536
    h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.05, 7)
    h.synth_rows(0, 999, 3_000_000, rows, NNZ)
    d, m, ids = submodel_minibatch(capi, oracle, h, 999, 3_000_000, rows, rows, 256, 1)
    h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, rows, 256)
    w, v = h.get_param_rows(ids)
    fid = d.entries["id"].reshape(rows, NNZ)                     # sub-model ids, one row of 32 per example
    counts = np.bincount(d.entries["id"], minlength=len(ids))
    free_example = (counts[fid] == 1).all(axis=1)                # every feature of the example occurs once in the launch
    assert 0.40 < free_example.mean() < 0.65                     # (1 - 0.0207)^32 = 0.51
    free = np.unique(fid[free_example])
    racing = np.setdiff1d(np.arange(len(ids)), free)
    np.testing.assert_allclose(v[:, free], m.v[:, free], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w[free], m.w[free], rtol=1e-4, atol=1e-6)
    # the rest: every racing writer applies a step of the right size from a value at most one update old
    assert np.abs(v[:, racing] - m.v[:, racing]).max() < 2e-2
    assert np.abs(v[:, racing] - m.v[:, racing]).mean() < 2e-5
    assert abs(h.get_w0() - m.w0) <= 2e-3                        # the bias recurrence sees the racing examples' sums (measured 4e-4)
    h.close()


def test_pairwise_term_is_homogeneous_of_degree_two(capi):
    """with w = 0 and w0 = 0 the prediction is the pairwise term only; scaling V by 2 scales it by 4"""
    rows = 1 << 18
    out = []
    for s in (0.02, 0.04):
        h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0, 0, 0, 0.01, -1.0, 1.0)
        h.init_params(0.0, s, 11)
        h.synth_rows(0, 9, 0, rows, NNZ)
        out.append(h.predict(0, rows))
        h.close()
    np.testing.assert_allclose(out[1], 4.0 * out[0], rtol=2e-5, atol=1e-7)


def test_segmented_minibatch_is_deterministic_at_full_size(capi):
    rows = 1 << 20
    sums = []
    for _ in range(2):
        h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
        h.init_params(0.0, 0.05, 3)
        h.synth_rows(0, 77, 0, rows, NNZ)
        h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 16384, 256, capi.FLAG_BIAS_LAG)
        p = h.predict(0, rows)
        sums.append((p.tobytes(), h.get_w0()))
        ev = h.evaluate(0)
        h.close()
    assert sums[0] == sums[1]                                  # bit-identical predictions and bias
    assert ev.accuracy > 0.5                                   # and the step learned something on its own rows


def test_feature_shards_sum_to_unsharded_partials_at_full_size(capi):
    import torch
    rows, world = 8192, 2
    full = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    full.init_params(0.0, 0.05, 5)
    full.synth_rows(0, 55, 0, rows, NNZ)
    nf = full.partial_floats(rows)
    ref = torch.zeros(nf, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    full.sgd_partial(0, 0, rows, ref.data_ptr())
    full.synchronize()
    tot = torch.zeros(nf, dtype=torch.float32, device="cuda")
    for r in range(world):
        s = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, shard_rank=r, shard_world=world)
        s.init_params(0.0, 0.05, 5)
        s.synth_rows(0, 55, 0, rows, NNZ)
        buf = torch.zeros(nf, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        s.sgd_partial(0, 0, rows, buf.data_ptr())
        s.synchronize()
        tot += buf
        s.close()
    torch.cuda.synchronize()
    np.testing.assert_allclose(tot.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=1e-6)
    full.close()


def test_largest_feature_ids(capi, oracle):
    """num_attribute at the reference's type limit (uint ids, fm_model.h:51): 2^32 - 1 features, rows touching the very
    last ids -- every index product must be 64-bit (k = 2: 34 GB of factors + 17 GB of linear weights)."""
    n = 2 ** 32 - 1
    k = 2
    h = capi.Handle(n, k, True, True, capi.TASK_REGRESSION, 0.0, 0.001, 0.002, 0.01, -5.0, 5.0)
    h.init_params(0.0, 0.1, 13)
    rng = np.random.default_rng(0)
    rows, nnz = 512, 6
    ids = np.concatenate([np.array([n - 1, n - 2, 0, 2 ** 31, 2 ** 31 + 1, 2 ** 32 - 7], dtype=np.uint64),
                          rng.integers(0, n, rows * nnz - 6, dtype=np.uint64)]).astype(np.uint32)
    ent = np.zeros(rows * nnz, dtype=capi.ENTRY_DTYPE)
    ent["id"] = ids
    ent["value"] = np.round(rng.uniform(0.5, 1.5, rows * nnz), 3)
    rp = np.arange(rows + 1, dtype=np.uint64) * np.uint64(nnz)
    y = rng.normal(0, 1, rows).astype(np.float32)
    h.upload_rows(0, ent, rp, y)
    uniq = np.unique(ids)
    w, v = h.get_param_rows(uniq)
    assert np.abs(v).max() > 0                                   # the fill reached the last rows
    m = oracle.Model(len(uniq), k, True, True, 0.0, 0.001, 0.002)
    m.w[:], m.v[:] = w, v
    e2 = ent.copy()
    e2["id"] = np.searchsorted(uniq, ids).astype(np.uint32)
    d = oracle.Data(e2, rp, y)
    np.testing.assert_allclose(h.predict(0, rows), oracle.predict_raw(m, d), rtol=1e-4, atol=2e-5)
    h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 64, 16)
    oracle.sgd_epoch_minibatch(m, d, 0, 0.01, -5.0, 5.0, 64, 16)
    w2, v2 = h.get_param_rows(uniq)
    np.testing.assert_allclose(v2, m.v, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w2, m.w, rtol=1e-4, atol=1e-6)
    with pytest.raises(capi.FmxError):
        capi.Handle(2 ** 32, 1)                                   # one past the reference's uint range
    h.close()
