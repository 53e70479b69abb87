"""CPU, world_size 2, gloo: the N>1 decomposition (feature shards + ONE all-reduce per minibatch) reproduces
the unsharded restated batch rule.  The kernels are GPU-only, so here the per-shard arithmetic is done by a
small numpy stand-in that follows libfm_amd/sharding.py's contract (same buffer layout, same ownership rule,
same order of steps as fmx_sgd_partial -> all_reduce -> fmx_sgd_finish); the result is compared with
oracle.sgd_epoch_minibatch on the full model.  What this pins: ownership/renumbering, the exchange layout,
that c and S are plain sums over features (so the all-reduce is the ONLY exchange), and that the multipliers
and the w0 recurrence computed redundantly on every rank agree."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, chunk, shard_hash, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from common import Golden
    from libfm_amd import sharding as sh
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = Golden("sgd_cls_zipf_k32")
    m = g.model(O, "init")
    tr = g.data(O, "train")
    k, n, KP = g.k, g.n, g.k                                  # k = 32 is already a power of two
    # this rank's shard: its rows of V / w and its view of the rows
    mine = sh.owned_ids(n, rank, world, shard_hash).astype(np.int64)   # global id of every local row (product code)
    V = m.v[:, mine].T.copy()                                 # [n_local][k]
    w = m.w[mine].copy()
    w0 = m.w0
    ent, rp = sh.filter_rows(tr.entries, tr.row_ptr, rank, world, n, shard_hash)
    rp = rp.astype(np.int64)
    y = tr.target.astype(np.float64)
    lr, (reg0, regw, regv) = g.lr, g.reg
    for _ in range(g.iters):
        for row0 in range(0, tr.n_rows, B):
            nb = min(B, tr.n_rows - row0)
            buf = np.zeros(sh.partial_floats(nb, KP), dtype=np.float64)
            S, c = sh.split_partial(buf, nb, KP)
            for e in range(nb):                               # fmx_sgd_partial
                a, b = rp[row0 + e], rp[row0 + e + 1]
                ids, x = ent["id"][a:b], ent["value"][a:b].astype(np.float64)
                d = V[ids] * x[:, None]
                S[e] = d.sum(0)
                c[e] = (w[ids] * x).sum() * g.k1 - 0.5 * (d * d).sum()
            t = torch.from_numpy(buf)
            dist.all_reduce(t)                                # the ONE exchange
            rest = c + 0.5 * (S * S).sum(1)                   # fmx_sgd_finish: identical on every rank
            mult = np.zeros(nb)
            for c0 in range(0, nb, chunk):
                nc = min(chunk, nb - c0)
                w0s = w0 if g.k0 else 0.0
                for e in range(c0, c0 + nc):
                    mult[e] = O.lib().fmo_multiplier(g.task, w0s + rest[e], y[row0 + e], g.min_target, g.max_target)
                if g.k0:
                    w0 -= lr * (mult[c0:c0 + nc].sum() + nc * reg0 * w0s)
            dV, dw = np.zeros_like(V), np.zeros_like(w)      # scatter-add into the LOCAL shard only
            for e in range(nb):
                a, b = rp[row0 + e], rp[row0 + e + 1]
                for i in range(a, b):
                    j, x = ent["id"][i], float(ent["value"][i])
                    dV[j] += -lr * (mult[e] * (S[e] * x - V[j] * x * x) + regv * V[j])
                    if g.k1:
                        dw[j] += -lr * (mult[e] * x + regw * w[j])
            V += dV
            w += dw
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), V=V, w=w, w0=w0, mine=mine)
    dist.destroy_process_group()


@pytest.mark.parametrize("B,chunk,shard_hash", [(100, 10, 1), (64, 64, 0)])
def test_two_feature_shards_equal_the_unsharded_rule(oracle, tmp_path, B, chunk, shard_hash):
    import torch.multiprocessing as mp
    from common import Golden
    world, port = 2, 29600 + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, B, chunk, shard_hash, str(tmp_path)), nprocs=world, join=True)
    g = Golden("sgd_cls_zipf_k32")
    m = g.model(oracle, "init")
    tr = g.data(oracle, "train")
    for _ in range(g.iters):
        oracle.sgd_epoch_minibatch(m, tr, g.task, g.lr, g.min_target, g.max_target, B, chunk)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_allclose(z["V"], m.v[:, z["mine"]].T, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(z["w"], m.w[z["mine"]], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(float(z["w0"]), m.w0, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("shard_hash", [0, 1])
def test_filter_rows_matches_ownership_rule(shard_hash):
    sys.path.insert(0, ROOT)
    from libfm_amd import sharding as sh
    import datagen
    ent, rp, _ = datagen.ragged_real(97, 50, 9, seed=3, empty_every=7)
    total = 0
    for r in range(3):
        e, p = sh.filter_rows(ent, rp, r, 3, 97, shard_hash)
        assert len(p) == len(rp) and p[-1] == len(e)
        total += len(e)
        assert (e["id"] < sh.n_local(97, r, 3)).all()
    assert total == len(ent)


@pytest.mark.parametrize("n,world", [(97, 3), (1000, 8), (65536, 8), (65537, 5), (1, 1), (7, 8), (3_000_001, 8)])
def test_hashed_ownership_is_a_balanced_bijection(n, world):
    """the library's hashed rule (fmx_shard_place / fmx_shard_global, include/fmx.h): every feature has one (owner, local
    row), local rows are dense, the inverse recovers the id, and structured id sets spread over the shards"""
    sys.path.insert(0, ROOT)
    from libfm_amd import sharding as sh
    ids = np.arange(n, dtype=np.uint32)
    own, loc = sh.place(ids, n, world, 1)
    assert own.min() >= 0 and own.max() < world
    key = loc.astype(np.int64) * world + own
    assert np.array_equal(np.sort(key), np.arange(n))                  # a permutation of [0, n): dense local tables
    for r in range(world):
        back = sh.owned_ids(n, r, world, 1)
        assert len(back) == sh.n_local(n, r, world) == int((own == r).sum())
        assert np.array_equal(own[back], np.full(len(back), r)) and np.array_equal(loc[back], np.arange(len(back)))
    if n >= 65536:
        for stride in (world, 2 * world, 1024):                        # ids of one residue class: all on one shard under mod
            cnt = np.bincount(own[::stride], minlength=world)
            assert cnt.max() < 1.25 * cnt.mean() + 8, (stride, cnt)
    own0, loc0 = sh.place(ids, n, world, 0)                            # the plain rule
    assert np.array_equal(own0, ids % world) and np.array_equal(loc0, ids // world)
