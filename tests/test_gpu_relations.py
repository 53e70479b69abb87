"""GPU: block-structured data (`-relation`).  FMX_BLOCKS_KEEP keeps main rows and blocks apart and sweeps the blocks through
per-block-row caches (the reference's algorithm, fm_learn_mcmc.h:478-527, 734-790, 849-909, restated); FMX_BLOCKS_EXPAND joins
the rows on the device.  Either way ALS must land on the REAL reference's block-structured run (fixtures rel_als_*).
Tolerance 1e-4 as for every ALS test."""
import contextlib
import io
import os

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

CASES = [c for c in golden_cases() if c.startswith("rel_als_")]


def blocks_of(z):
    return [(z["rel%d_entries" % i], z["rel%d_row_ptr" % i], int(z["rel%d_num_feature" % i])) for i in range(int(z["n_relations"]))]


def test_device_expansion_is_the_flat_design_matrix():
    from libfm_amd import capi
    (ent, rp, y), blocks, maps = datagen.block_structured(40, 25, 300, seed=5)
    n_main = 7
    flat_ent, flat_rp, offs = datagen.expand_blocks(ent, rp, blocks, maps, n_main)
    n = offs[-1] + blocks[-1][2]
    h = capi.Handle(n, 4)
    h.upload_block_rows(0, ent, rp, y, [(be, bp, mp, off) for (be, bp, _), mp, off in zip(blocks, maps, offs)])
    got_ent, got_rp, got_y = h.download_rows(0)
    assert len(got_y) == 300 and len(got_ent) == len(flat_ent)
    assert np.array_equal(got_rp, flat_rp) and np.array_equal(got_ent, flat_ent) and np.array_equal(got_y, y)
    # malformed mapping -> loud error
    bad = maps[0].copy(); bad[3] = 10 ** 6
    with pytest.raises(capi.FmxError):
        h.upload_block_rows(1, ent, rp, y, [(blocks[0][0], blocks[0][1], bad, offs[0])])
    h.close()


def test_kept_blocks_predict_like_the_expanded_rows(oracle):
    """FMX_BLOCKS_KEEP: fm_model::predict / evaluate add the block rows' sums through the mappings -- same numbers as on the
    materialised join; the SGD learners refuse such rows like the reference (fm_learn_sgd.h:61-63)."""
    from libfm_amd import capi
    (ent, rp, y), blocks, maps = datagen.block_structured(40, 25, 300, seed=5)
    n_main = 7
    flat_ent, flat_rp, offs = datagen.expand_blocks(ent, rp, blocks, maps, n_main)
    n, k = offs[-1] + blocks[-1][2], 8
    m = oracle.Model(n, k, True, True, 0.0, 0.0, 0.0)
    m.v[:] = oracle.init_values(3, n, k, 0.3)
    m.w[:] = oracle.init_values(4, n, 1, 0.3)[0]
    m.w0 = 0.25
    h = capi.Handle(n, k, True, True, 0, 0, 0, 0, 0.01, float(y.min()), float(y.max()))
    h.set_params(m.w0, m.w, m.v)
    rel = [(be, bp, mp, off) for (be, bp, _), mp, off in zip(blocks, maps, offs)]
    h.upload_block_rows(0, ent, rp, y, rel, keep=True)
    h.upload_block_rows(1, ent, rp, y, rel, keep=False)
    want = oracle.predict_raw(m, oracle.Data(flat_ent, flat_rp, y))
    np.testing.assert_allclose(h.predict(0, 300), want, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(h.predict(1, 300), want, rtol=1e-4, atol=5e-5)
    assert abs(h.evaluate(0).rmse - h.evaluate(1).rmse) < 1e-5
    with pytest.raises(capi.FmxError, match="relations are not supported with SGD"):
        h.sgd_epoch(0, capi.SGD_MINIBATCH)
    h.close()


@pytest.mark.parametrize("keep,split", [(True, False), (False, False), (True, True), (False, True)])
@pytest.mark.parametrize("name", CASES)
def test_als_on_relations_matches_reference(oracle, name, keep, split, monkeypatch):
    _split(monkeypatch, "1" if split else "0")   # main features: split / fused form of the draws
    from libfm_amd import data as D
    from libfm_amd import learner as L
    g = Golden(name)
    z = g.z
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = g.n, g.k, bool(g.k0), bool(g.k1)
    fm.reg0, fm.regw, fm.regv = g.reg
    m = g.model(oracle, "init")
    fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
    l = L.FMLearnALS()
    l.fm, l.task, l.num_iter, l.min_target, l.max_target = fm, g.task, g.iters, g.min_target, g.max_target
    l.w_lambda, l.v_lambda = g.reg[1], g.reg[2]
    if "group" in z.files:
        l.groups, l.w_lambda, l.v_lambda = z["group"], z["w_lambda_g"], z["v_lambda_g"]
    l.out = io.StringIO()
    train = L.Data(z["train_entries"], z["train_row_ptr"], g.train_target)
    test = L.Data(z["test_entries"], z["test_row_ptr"], g.test_target)
    off = int(z["n_main"])
    for i, (be, bp, nf) in enumerate(blocks_of(z)):
        rel = D.Relation(be, bp, nf)
        train.add_relation(rel, z["rel%d_train" % i], off)
        test.add_relation(rel, z["rel%d_test" % i], off)
        off += nf
    train.keep_blocks = test.keep_blocks = keep                # per-block caches (like the reference) / materialised join
    l.init()
    l.learn(train, test)
    assert abs(l.fm.w0 - float(z["final_w0"])) <= 1e-4 * abs(float(z["final_w0"])) + 2e-5
    np.testing.assert_allclose(l.fm.w, z["final_w"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.fm.v, z["final_v"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l.predict(test), z["pred_out"], rtol=1e-4, atol=5e-5)
    l.close()


@pytest.mark.parametrize("name", CASES)
def test_sampled_chain_on_kept_blocks_is_the_chain_on_joined_rows(oracle, name):
    """do_sample = 1 (Gibbs draws, fixed hyper-parameters): the noise of a coordinate is keyed by (seed, iteration, family,
    GLOBAL attribute id), the probit targets by the data row -- so the block-wise sweep must draw the chain of the joined rows."""
    from libfm_amd import data as D
    from libfm_amd import learner as L
    g = Golden(name)
    z = g.z
    res = []
    for keep in (True, False):
        fm = L.FMModel()
        fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = g.n, g.k, bool(g.k0), bool(g.k1)
        m = g.model(oracle, "init")
        fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
        l = L.FMLearnALS()
        l.fm, l.task, l.num_iter, l.min_target, l.max_target = fm, g.task, 6, g.min_target, g.max_target
        l.w_lambda, l.v_lambda, l.do_sample, l.seed = 2.0, 3.0, True, 1234
        l.out = io.StringIO()
        train = L.Data(z["train_entries"], z["train_row_ptr"], g.train_target)
        test = L.Data(z["test_entries"], z["test_row_ptr"], g.test_target)
        off = int(z["n_main"])
        for i, (be, bp, nf) in enumerate(blocks_of(z)):
            rel = D.Relation(be, bp, nf)
            train.add_relation(rel, z["rel%d_train" % i], off)
            test.add_relation(rel, z["rel%d_test" % i], off)
            off += nf
        train.keep_blocks = test.keep_blocks = keep
        l.init()
        l.learn(train, test)
        res.append((l.fm.w0, l.fm.w.copy(), l.fm.v.copy(), l.predict(test).copy()))
        l.close()
    (w0a, wa, va, pa), (w0b, wb, vb, pb) = res
    assert np.abs(va).max() > 0.05 and np.abs(va - g.model(oracle, "init").v).max() > 0.05      # the chain moved
    assert abs(w0a - w0b) <= 1e-4 * abs(w0b) + 2e-5
    np.testing.assert_allclose(wa, wb, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(va, vb, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(pa, pb, rtol=1e-4, atol=1e-4)


def test_cli_relation_flag(tmp_path, oracle):
    """`-method als -relation a,b` with the reference's file set (<a>.xt, <a>.train, <a>.test, <a>.groups) and the same
    seed as the reference run behind the fixture -> the same -out predictions."""
    from libfm_amd import cli
    from libfm_amd import data as D
    g = Golden("rel_als_cls_groups")
    z = g.z
    trf, tef = str(tmp_path / "tr.libfm"), str(tmp_path / "te.libfm")
    oracle.Data(z["train_entries"], z["train_row_ptr"], z["train_target"]).write_libsvm(trf)
    oracle.Data(z["test_entries"], z["test_row_ptr"], z["test_target"]).write_libsvm(tef)
    names = []
    for i, (be, bp, nf) in enumerate(blocks_of(z)):
        px = str(tmp_path / ("rel%d" % i))
        et, cp = D.transpose(be, bp, nf)
        D.write_binary_matrix(px + ".xt", et, cp, num_cols=len(bp) - 1)
        np.savetxt(px + ".train", z["rel%d_train" % i], fmt="%d")
        np.savetxt(px + ".test", z["rel%d_test" % i], fmt="%d")
        np.savetxt(px + ".groups", z["rel%d_groups" % i], fmt="%d")
        names.append(px)
    reg = [float(z["reg"][0])] + [float(x) for x in z["w_lambda_g"]] + [float(x) for x in z["v_lambda_g"]]
    out = str(tmp_path / "pred")
    argv = ["-task", "c", "-train", trf, "-test", tef, "-dim", "%d,%d,%d" % (g.k0, g.k1, g.k), "-iter", str(g.iters),
            "-method", "als", "-relation", ",".join(names), "-regular", ",".join(repr(x) for x in reg),
            "-init_stdev", repr(float(z["init_stdev"])), "-seed", str(int(z["seed"])), "-out", out]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(argv) == 0
    assert "#relations: 2" in buf.getvalue() and "#groups=5" in buf.getvalue()
    np.testing.assert_allclose(np.loadtxt(out), z["pred_out"], rtol=1e-4, atol=5e-5)
    assert os.path.exists(out)


@pytest.mark.parametrize("world,shard_hash", [(2, 1), (3, 0)])
def test_joined_blocks_on_feature_shards(oracle, world, shard_hash):
    """`-relation` with gpu_devices (round-2 advisor finding: the upload used to fail on every shard).  FMX_BLOCKS_EXPAND on a feature
    shard joins the rows on the host and keeps the shard's own features: every shard holds exactly what it holds after an upload of
    the flat design matrix, a sharded ALS sweep over them is the unsharded sweep, and kept blocks are refused with a clear text."""
    from libfm_amd import capi
    (ent, rp, y), blocks, maps = datagen.block_structured(40, 25, 300, seed=5)
    n_main = 7
    flat_ent, flat_rp, offs = datagen.expand_blocks(ent, rp, blocks, maps, n_main)
    n, k = offs[-1] + blocks[-1][2], 8
    rel = [(be, bp, mp, off) for (be, bp, _), mp, off in zip(blocks, maps, offs)]
    m = oracle.Model(n, k, True, True, 0.1, 1.0, 5.0)
    m.v[:] = oracle.init_values(3, n, k, 0.1)
    m.w[:] = oracle.init_values(4, n, 1, 0.1)[0]
    lo, hi = float(y.min()), float(y.max())
    hs = [capi.Handle(n, k, True, True, 0, 0.1, 1.0, 5.0, 0.0, lo, hi, device=0, shard_rank=r, shard_world=world, shard_hash=shard_hash)
          for r in range(world)]
    ref = [capi.Handle(n, k, True, True, 0, 0.1, 1.0, 5.0, 0.0, lo, hi, device=0, shard_rank=r, shard_world=world, shard_hash=shard_hash)
           for r in range(world)]
    for h, q in zip(hs, ref):
        h.set_params(m.w0, m.w, m.v)
        h.upload_block_rows(0, ent, rp, y, rel, keep=False)
        q.upload_rows(0, flat_ent, flat_rp, y)
        a, b = h.download_rows(0), q.download_rows(0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        with pytest.raises(capi.FmxError, match="FMX_BLOCKS_EXPAND"):
            h.upload_block_rows(1, ent, rp, y, rel, keep=True)
    g = capi.Group(hs)
    g.als_begin(0)
    g.als_sweep(1.0, 5.0)
    g.als_end()
    one = capi.Handle(n, k, True, True, 0, 0.1, 1.0, 5.0, 0.0, lo, hi, device=0)
    one.set_params(m.w0, m.w, m.v)
    one.upload_block_rows(0, ent, rp, y, rel, keep=False)
    one.als_begin(0)
    one.als_sweep(1.0, 5.0)
    one.als_end()
    w0, w, v = g.get_params()
    w0b, wb, vb = one.get_params()
    np.testing.assert_allclose(v, vb, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(w, wb, rtol=1e-4, atol=2e-5)
    assert abs(w0 - w0b) <= 1e-4 * abs(w0b) + 2e-5
    g.close()
    for h in hs + ref + [one]:
        h.close()
