#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL reference.

Needs /root/reference (build container only): `make -C oracle` compiles oracle/_ref/ref_harness from the
reference sources where they lie; this script feeds it seeded data sets (tests/datagen.py) and stores the
inputs, the configuration and the reference's full-precision outputs.  The fixtures pin the C restatement
(oracle/fm_oracle.c) in the `-m "not gpu"` tests and the HIP path in the `-m gpu` tests; the GPU box has no
/root/reference, so the committed .npz files are what travels.

    python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import datagen  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    # name: (generator, kwargs_train, kwargs_test, config)
    "sgd_reg_ml": dict(gen="movielens_shaped", train=dict(n_users=120, n_items=80, n_rows=600, seed=11),
                       test=dict(n_users=120, n_items=80, n_rows=150, seed=12),
                       cfg=dict(task="r", k0=1, k1=1, k=8, iters=5, lr=0.01, reg=(0.0, 0.0, 0.01), init_stdev=0.1, seed=42)),
    "sgd_cls_ragged": dict(gen="ragged_real", train=dict(n_features=300, n_rows=400, max_nnz=12, seed=21, empty_every=50),
                           test=dict(n_features=300, n_rows=100, max_nnz=12, seed=22),
                           cfg=dict(task="c", k0=1, k1=1, k=16, iters=4, lr=0.02, reg=(0.001, 0.002, 0.003), init_stdev=0.05, seed=7)),
    "sgd_reg_ragged_nolin": dict(gen="ragged_real", train=dict(n_features=200, n_rows=300, max_nnz=9, seed=31, classification=False),
                                 test=dict(n_features=200, n_rows=80, max_nnz=9, seed=32, classification=False),
                                 cfg=dict(task="r", k0=0, k1=0, k=4, iters=3, lr=0.01, reg=(0.0, 0.0, 0.02), init_stdev=0.1, seed=3)),
    "sgd_cls_k64": dict(gen="onehot_fields", train=dict(n_features=640, nnz=8, n_rows=300, seed=41),
                        test=dict(n_features=640, nnz=8, n_rows=100, seed=42),
                        cfg=dict(task="c", k0=1, k1=1, k=64, iters=3, lr=0.01, reg=(0.0, 0.0, 0.001), init_stdev=0.01, seed=1)),
    "sgd_cls_dup": dict(gen="ragged_real", train=dict(n_features=60, n_rows=120, max_nnz=6, seed=51, duplicates=True),
                        test=dict(n_features=60, n_rows=40, max_nnz=6, seed=52),
                        cfg=dict(task="c", k0=1, k1=1, k=8, iters=2, lr=0.05, reg=(0.0, 0.01, 0.01), init_stdev=0.1, seed=5)),
    "sgd_reg_k1": dict(gen="movielens_shaped", train=dict(n_users=30, n_items=20, n_rows=200, seed=61),
                       test=dict(n_users=30, n_items=20, n_rows=50, seed=62),
                       cfg=dict(task="r", k0=1, k1=1, k=1, iters=3, lr=0.02, reg=(0.0, 0.0, 0.0), init_stdev=0.1, seed=9)),
    "sgd_cls_zipf_k32": dict(gen="onehot_fields", train=dict(n_features=1600, nnz=16, n_rows=500, seed=71, zipf=1.05),
                             test=dict(n_features=1600, nnz=16, n_rows=100, seed=72, zipf=1.05),
                             cfg=dict(task="c", k0=1, k1=1, k=32, iters=2, lr=0.01, reg=(0.0, 0.0, 0.001), init_stdev=0.01, seed=2)),
}


def groups_of(case, n):
    """attribute -> group array of a case: ('split', boundary) = ids below the boundary are group 0, the rest group 1
    (users / items); ('fields', nnz, n_features, fields_per_group) = one-hot field blocks."""
    spec = case.get("groups")
    if spec is None:
        return None
    j = np.arange(n)
    if spec[0] == "split":
        return (j >= spec[1]).astype(np.uint32)
    _, nnz, n_features, per = spec
    return np.minimum(j // (n_features // nnz), nnz - 1).astype(np.uint32) // per


def write_meta(path, case, n_nominal):
    g = groups_of(case, n_nominal)
    with open(path, "w") as f:                       # DVector<uint>::load reads `dim` whitespace-separated values (matrix.h:360-371)
        f.write("".join("%d\n" % x for x in g))


ALS_CASES = {
    "als_reg_ml": dict(gen="movielens_shaped", train=dict(n_users=120, n_items=80, n_rows=600, seed=11),
                       test=dict(n_users=120, n_items=80, n_rows=150, seed=12),
                       cfg=dict(task="r", k0=1, k1=1, k=8, iters=4, reg=(0.0, 1.0, 5.0), init_stdev=0.1, seed=42)),
    "als_cls_ragged": dict(gen="ragged_real", train=dict(n_features=150, n_rows=300, max_nnz=8, seed=21, empty_every=50),
                           test=dict(n_features=160, n_rows=80, max_nnz=8, seed=22),
                           cfg=dict(task="c", k0=1, k1=1, k=4, iters=3, reg=(0.5, 2.0, 8.0), init_stdev=0.05, seed=7)),
    "als_reg_fields_k16": dict(gen="onehot_fields", train=dict(n_features=480, nnz=6, n_rows=400, seed=41, classification=False),
                               test=dict(n_features=480, nnz=6, n_rows=100, seed=42, classification=False),
                               cfg=dict(task="r", k0=1, k1=1, k=16, iters=3, reg=(0.0, 0.5, 10.0), init_stdev=0.1, seed=1)),
    # attribute groups (-meta) with per-group lambdas (-regular 'r0,w_1..w_G,v_1..v_G', libfm.cpp:353-363)
    "als_reg_ml_groups": dict(gen="movielens_shaped", train=dict(n_users=120, n_items=80, n_rows=600, seed=11),
                              test=dict(n_users=120, n_items=80, n_rows=150, seed=12), groups=("split", 120), n_nominal=200,
                              group_reg=((0.5, 2.0), (3.0, 9.0)),
                              cfg=dict(task="r", k0=1, k1=1, k=8, iters=4, reg=(0.0, 0.0, 0.0), init_stdev=0.1, seed=42)),
    "als_cls_fields_groups": dict(gen="onehot_fields", train=dict(n_features=480, nnz=6, n_rows=400, seed=43),
                                  test=dict(n_features=480, nnz=6, n_rows=100, seed=44), groups=("fields", 6, 480, 2), n_nominal=480,
                                  group_reg=((0.1, 1.0, 4.0), (2.0, 6.0, 12.0)),
                                  cfg=dict(task="c", k0=1, k1=1, k=4, iters=3, reg=(0.2, 0.0, 0.0), init_stdev=0.1, seed=8)),
    "als_reg_nolin_dup": dict(gen="ragged_real", train=dict(n_features=60, n_rows=150, max_nnz=6, seed=51, duplicates=True, classification=False),
                              test=dict(n_features=60, n_rows=40, max_nnz=6, seed=52, classification=False),
                              cfg=dict(task="r", k0=0, k1=0, k=3, iters=2, reg=(0.0, 0.0, 2.0), init_stdev=0.1, seed=5)),
}


def make_als():
    for name, case in ALS_CASES.items():
        gen = getattr(datagen, case["gen"])
        tr = O.Data(*gen(**case["train"]))
        te = O.Data(*gen(**case["test"]))
        cfg = case["cfg"]
        with tempfile.TemporaryDirectory() as td:
            trf, tef, pre = os.path.join(td, "train.libfm"), os.path.join(td, "test.libfm"), os.path.join(td, "out")
            tr.write_libsvm(trf)
            te.write_libsvm(tef)
            env, extra = {}, {}
            if "groups" in case:
                write_meta(os.path.join(td, "meta"), case, case["n_nominal"])
                env = {"FMX_META": os.path.join(td, "meta"),
                       "FMX_GROUP_REG": ",".join(repr(x) for x in case["group_reg"][0] + case["group_reg"][1])}
            O.run_ref_harness(["als", trf, tef, cfg["task"], cfg["k0"], cfg["k1"], cfg["k"], cfg["iters"],
                               repr(cfg["reg"][0]), repr(cfg["reg"][1]), repr(cfg["reg"][2]), repr(cfg["init_stdev"]),
                               cfg["seed"], pre], env=env)
            init = O.Model.from_dump(pre + ".init.bin")
            final = O.Model.from_dump(pre + ".final.bin")
            pred_out = np.fromfile(pre + ".pred_out.bin", dtype=np.float64)
            if "groups" in case:
                extra = dict(group=groups_of(case, init.n), w_lambda_g=np.array(case["group_reg"][0]),
                             v_lambda_g=np.array(case["group_reg"][1]))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), **extra,
            train_entries=tr.entries, train_row_ptr=tr.row_ptr, train_target=tr.target,
            test_entries=te.entries, test_row_ptr=te.row_ptr, test_target=te.target,
            task=cfg["task"], k0=cfg["k0"], k1=cfg["k1"], k=cfg["k"], iters=cfg["iters"], lr=0.0,
            reg=np.array(cfg["reg"]), init_stdev=cfg["init_stdev"], seed=cfg["seed"],
            n=init.n, init_w0=init.w0, init_w=init.w, init_v=init.v,
            final_w0=final.w0, final_w=final.w, final_v=final.v, pred_out=pred_out)
        print("%-22s n=%d k=%d rows=%d/%d  pred_out[:3]=%s" % (name, init.n, init.k, tr.n_rows, te.n_rows, pred_out[:3]))


MCMC_CASES = {
    "mcmc_reg_ml": dict(gen="movielens_shaped", train=dict(n_users=300, n_items=200, n_rows=6000, seed=11),
                        test=dict(n_users=300, n_items=200, n_rows=1500, seed=11, _skip=6000),
                        cfg=dict(task="r", k0=1, k1=1, k=8, iters=40, init_stdev=0.1, seed=42)),
    "mcmc_cls_fields": dict(gen="onehot_fields", train=dict(n_features=600, nnz=6, n_rows=4000, seed=41),
                            test=dict(n_features=600, nnz=6, n_rows=1000, seed=41, _skip=4000),
                            cfg=dict(task="c", k0=1, k1=1, k=4, iters=40, init_stdev=0.1, seed=1)),
    # round 4: the bench's factor count (the bands above are k = 4 / 8): generated and banded on its own, `--mcmc-only mcmc_reg_ml_k64`
    "mcmc_reg_ml_k64": dict(gen="movielens_shaped", train=dict(n_users=300, n_items=200, n_rows=6000, seed=17),
                            test=dict(n_users=300, n_items=200, n_rows=1500, seed=17, _skip=6000),
                            cfg=dict(task="r", k0=1, k1=1, k=64, iters=40, init_stdev=0.1, seed=42)),
    # round 5: BASELINE configs[4]'s factor count (k = 128: two floats per lane, the VEC = 2 path of the draws WITH noise), `--mcmc-only mcmc_reg_ml_k128`
    "mcmc_reg_ml_k128": dict(gen="movielens_shaped", train=dict(n_users=300, n_items=200, n_rows=6000, seed=19),
                             test=dict(n_users=300, n_items=200, n_rows=1500, seed=19, _skip=6000),
                             cfg=dict(task="r", k0=1, k1=1, k=128, iters=40, init_stdev=0.1, seed=42)),
    "mcmc_reg_ml_groups": dict(gen="movielens_shaped", train=dict(n_users=300, n_items=200, n_rows=6000, seed=13),
                               test=dict(n_users=300, n_items=200, n_rows=1500, seed=13, _skip=6000), groups=("split", 300), n_nominal=500,
                               cfg=dict(task="r", k0=1, k1=1, k=8, iters=40, init_stdev=0.1, seed=42)),
}


def make_mcmc(only=None):
    """statistical fixtures: the reference's MCMC (libc rand() stream) on a train/test split of ONE planted model."""
    for name, case in MCMC_CASES.items():
        if only and name not in only:
            continue
        if not only and name in ("mcmc_reg_ml_k64", "mcmc_reg_ml_k128"):
            continue                                          # (added in rounds 4 / 5 without touching the round-1 fixtures)
        gen = getattr(datagen, case["gen"])
        kw = dict(case["train"])
        n_tr, n_te = kw["n_rows"], case["test"]["n_rows"]
        kw["n_rows"] = n_tr + n_te
        ent, rp, y = gen(**kw)
        rp = rp.astype(np.int64)
        tr = O.Data(ent[:rp[n_tr]], rp[:n_tr + 1], y[:n_tr])
        te = O.Data(ent[rp[n_tr]:], rp[n_tr:] - rp[n_tr], y[n_tr:])
        cfg = case["cfg"]
        with tempfile.TemporaryDirectory() as td:
            trf, tef, pre = os.path.join(td, "train.libfm"), os.path.join(td, "test.libfm"), os.path.join(td, "out")
            tr.write_libsvm(trf)
            te.write_libsvm(tef)
            env, extra = {}, {}
            if "groups" in case:
                write_meta(os.path.join(td, "meta"), case, case["n_nominal"])
                env = {"FMX_META": os.path.join(td, "meta")}
            O.run_ref_harness(["mcmc", trf, tef, cfg["task"], cfg["k0"], cfg["k1"], cfg["k"], cfg["iters"],
                               repr(cfg["init_stdev"]), cfg["seed"], pre], env=env)
            init = O.Model.from_dump(pre + ".init.bin")
            pred_out = np.fromfile(pre + ".pred_out.bin", dtype=np.float64)
            if "groups" in case:
                extra = dict(group=groups_of(case, init.n))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), **extra,
            train_entries=tr.entries, train_row_ptr=tr.row_ptr, train_target=tr.target,
            test_entries=te.entries, test_row_ptr=te.row_ptr, test_target=te.target,
            task=cfg["task"], k0=cfg["k0"], k1=cfg["k1"], k=cfg["k"], iters=cfg["iters"], lr=0.0,
            reg=np.zeros(3), init_stdev=cfg["init_stdev"], seed=cfg["seed"],
            n=init.n, init_w0=init.w0, init_w=init.w, init_v=init.v, pred_out=pred_out)
        if cfg["task"] == "r":
            print("%-22s n=%d rows=%d/%d test rmse of the reference's posterior mean: %.4f" % (name, init.n, tr.n_rows, te.n_rows, np.sqrt(np.mean((pred_out - te.target) ** 2))))
        else:
            yy = np.where(te.target > 0, 1, 0)
            print("%-22s n=%d rows=%d/%d test accuracy of the reference's posterior mean: %.4f" % (name, init.n, tr.n_rows, te.n_rows, np.mean((pred_out >= 0.5) == (yy == 1))))


SGDA_CASES = {
    "sgda_reg_ml": dict(gen="movielens_shaped", n=dict(n_users=120, n_items=80, seed=11), rows=(600, 150, 200),
                        cfg=dict(task="r", k0=1, k1=1, k=8, iters=4, lr=0.005, init_stdev=0.1, seed=42)),
    "sgda_cls_fields": dict(gen="onehot_fields", n=dict(n_features=300, nnz=6, seed=41), rows=(500, 100, 150),
                            cfg=dict(task="c", k0=1, k1=1, k=4, iters=3, lr=0.02, init_stdev=0.1, seed=3)),
    "sgda_reg_ml_groups": dict(gen="movielens_shaped", n=dict(n_users=120, n_items=80, seed=15), rows=(600, 150, 200),
                               groups=("split", 120), n_nominal=200,
                               cfg=dict(task="r", k0=1, k1=1, k=8, iters=4, lr=0.005, init_stdev=0.1, seed=42)),
    "sgda_cls_fields_groups": dict(gen="onehot_fields", n=dict(n_features=300, nnz=6, seed=45), rows=(500, 100, 150),
                                   groups=("fields", 6, 300, 1), n_nominal=300,
                                   cfg=dict(task="c", k0=1, k1=1, k=4, iters=3, lr=0.02, init_stdev=0.1, seed=3)),
}


def make_sgda():
    for name, case in SGDA_CASES.items():
        gen = getattr(datagen, case["gen"])
        ntr, nte, nva = case["rows"]
        ent, rp, y = gen(n_rows=ntr + nte + nva, **case["n"])
        rp = rp.astype(np.int64)

        def part(a, b):
            return O.Data(ent[rp[a]:rp[b]], rp[a:b + 1] - rp[a], y[a:b])
        tr, te, va = part(0, ntr), part(ntr, ntr + nte), part(ntr + nte, ntr + nte + nva)
        cfg = case["cfg"]
        with tempfile.TemporaryDirectory() as td:
            f = [os.path.join(td, x) for x in ("train", "test", "val")]
            for d, p in zip((tr, te, va), f):
                d.write_libsvm(p)
            pre = os.path.join(td, "out")
            env, extra = {}, {}
            if "groups" in case:
                write_meta(os.path.join(td, "meta"), case, case["n_nominal"])
                env = {"FMX_META": os.path.join(td, "meta")}
            O.run_ref_harness(["sgda", f[0], f[1], cfg["task"], cfg["k0"], cfg["k1"], cfg["k"], cfg["iters"], repr(cfg["lr"]),
                               "0", "0", "0", repr(cfg["init_stdev"]), cfg["seed"], pre, f[2]], env=env)
            init = O.Model.from_dump(pre + ".init.bin")
            final = O.Model.from_dump(pre + ".final.bin")
            pred_out = np.fromfile(pre + ".pred_out.bin", dtype=np.float64)
            regs = np.loadtxt(pre + ".reg.txt")
            ev = np.loadtxt(pre + ".eval.txt", ndmin=2)
            if "groups" in case:
                extra = dict(group=groups_of(case, init.n))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), **extra,
            train_entries=tr.entries, train_row_ptr=tr.row_ptr, train_target=tr.target,
            test_entries=te.entries, test_row_ptr=te.row_ptr, test_target=te.target,
            val_entries=va.entries, val_row_ptr=va.row_ptr, val_target=va.target,
            task=cfg["task"], k0=cfg["k0"], k1=cfg["k1"], k=cfg["k"], iters=cfg["iters"], lr=cfg["lr"],
            reg=np.zeros(3), init_stdev=cfg["init_stdev"], seed=cfg["seed"],
            n=init.n, init_w0=init.w0, init_w=init.w, init_v=init.v,
            final_w0=final.w0, final_w=final.w, final_v=final.v, pred_out=pred_out, regs=regs, eval=ev)
        print("%-22s n=%d k=%d rows=%d/%d/%d reg_w=%.4g reg_v[0]=%.4g" % (name, init.n, init.k, ntr, nte, nva, regs[0], regs[1]))


REL_CASES = {
    # block-structured data (`-relation`, relation.h): users block + items block, main table = one context attribute
    "rel_als_reg": dict(gen=dict(n_users=60, n_items=40, n_rows=700, seed=81), split=550, rel_groups=False,
                        cfg=dict(task="r", k0=1, k1=1, k=4, iters=4, reg=(0.0, 1.0, 6.0), init_stdev=0.1, seed=11)),
    # ... with <prefix>.groups files: joined meta = main(1) + users(2: ids | attributes) + items(2: ids | genres)
    "rel_als_cls_groups": dict(gen=dict(n_users=50, n_items=30, n_rows=600, seed=83, classification=True), split=480, rel_groups=True,
                               group_reg=((0.5, 1.0, 2.0, 1.5, 3.0), (4.0, 6.0, 9.0, 5.0, 12.0)),
                               cfg=dict(task="c", k0=1, k1=1, k=3, iters=3, reg=(0.1, 0.0, 0.0), init_stdev=0.1, seed=12)),
}


def make_rel():
    from libfm_amd import data as D                            # host-side file formats only (no GPU involved)
    for name, case in REL_CASES.items():
        (ent, rp, y), blocks, maps = datagen.block_structured(**case["gen"])
        rp = rp.astype(np.int64)
        ntr = case["split"]
        tr = O.Data(ent[:rp[ntr]], rp[:ntr + 1], y[:ntr])
        te = O.Data(ent[rp[ntr]:], rp[ntr:] - rp[ntr], y[ntr:])
        cfg = case["cfg"]
        n_main = 7
        with tempfile.TemporaryDirectory() as td:
            trf, tef, pre = os.path.join(td, "train.libfm"), os.path.join(td, "test.libfm"), os.path.join(td, "out")
            tr.write_libsvm(trf)
            te.write_libsvm(tef)
            prefixes, rgroups = [], []
            for bi, ((be, bp, nf), mp) in enumerate(zip(blocks, maps)):
                px = os.path.join(td, "rel%d" % bi)
                et, cp = D.transpose(be, bp, nf)
                D.write_binary_matrix(px + ".xt", et, cp, num_cols=len(bp) - 1)     # als/mcmc read only .xt (libfm.cpp:181-185)
                np.savetxt(px + ".train", mp[:ntr], fmt="%d")
                np.savetxt(px + ".test", mp[ntr:], fmt="%d")
                g = None
                if case["rel_groups"]:
                    n_ids = nf - (2 if bi == 0 else 3)
                    g = (np.arange(nf) >= n_ids).astype(np.uint32)
                    np.savetxt(px + ".groups", g, fmt="%d")
                rgroups.append(g)
                prefixes.append(px)
            env = {"FMX_RELATIONS": ",".join(prefixes)}
            if "group_reg" in case:
                env["FMX_GROUP_REG"] = ",".join(repr(x) for x in case["group_reg"][0] + case["group_reg"][1])
            O.run_ref_harness(["als", trf, tef, cfg["task"], cfg["k0"], cfg["k1"], cfg["k"], cfg["iters"],
                               repr(cfg["reg"][0]), repr(cfg["reg"][1]), repr(cfg["reg"][2]), repr(cfg["init_stdev"]),
                               cfg["seed"], pre], env=env)
            init = O.Model.from_dump(pre + ".init.bin")
            final = O.Model.from_dump(pre + ".final.bin")
            pred_out = np.fromfile(pre + ".pred_out.bin", dtype=np.float64)
        extra = {}
        for bi, ((be, bp, nf), mp) in enumerate(zip(blocks, maps)):
            extra.update({"rel%d_entries" % bi: be, "rel%d_row_ptr" % bi: bp, "rel%d_num_feature" % bi: nf,
                          "rel%d_train" % bi: mp[:ntr], "rel%d_test" % bi: mp[ntr:]})
            if rgroups[bi] is not None:
                extra["rel%d_groups" % bi] = rgroups[bi]
        if "group_reg" in case:                                  # joined meta (libfm.cpp:217-240): main group 0, then the blocks' groups
            grp, nxt = [np.zeros(n_main, dtype=np.uint32)], 1
            for g in rgroups:
                grp.append(g + nxt)
                nxt += int(g.max()) + 1
            extra.update(group=np.concatenate(grp).astype(np.uint32), w_lambda_g=np.array(case["group_reg"][0]),
                         v_lambda_g=np.array(case["group_reg"][1]))
        assert init.n == n_main + sum(nf for _, _, nf in blocks), (init.n, n_main)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), **extra, n_relations=len(blocks), n_main=n_main,
            train_entries=tr.entries, train_row_ptr=tr.row_ptr, train_target=tr.target,
            test_entries=te.entries, test_row_ptr=te.row_ptr, test_target=te.target,
            task=cfg["task"], k0=cfg["k0"], k1=cfg["k1"], k=cfg["k"], iters=cfg["iters"], lr=0.0,
            reg=np.array(cfg["reg"]), init_stdev=cfg["init_stdev"], seed=cfg["seed"],
            n=init.n, init_w0=init.w0, init_w=init.w, init_v=init.v,
            final_w0=final.w0, final_w=final.w, final_v=final.v, pred_out=pred_out)
        print("%-22s n=%d k=%d rows=%d/%d relations=%d pred_out[:3]=%s" % (name, init.n, init.k, tr.n_rows, te.n_rows, len(blocks), pred_out[:3]))


def make_c1():
    """BASELINE.json configs[0]: MovieLens-100K-shaped plumbing case run through the STOCK reference binary
    (oracle/_ref/libFM, the reference's own main + CLI + text parser + -out / -save_model writers)."""
    import subprocess
    ent, rp, y = datagen.movielens_shaped(943, 1682, 100000, seed=100)
    rp = rp.astype(np.int64)
    ntr = 80000
    tr = O.Data(ent[:rp[ntr]], rp[:ntr + 1], y[:ntr])
    te = O.Data(ent[rp[ntr]:], rp[ntr:] - rp[ntr], y[ntr:])
    with tempfile.TemporaryDirectory() as td:
        trf, tef = os.path.join(td, "ml.train.libfm"), os.path.join(td, "ml.test.libfm")
        tr.write_libsvm(trf)
        te.write_libsvm(tef)
        cmd = [O.REF_LIBFM, "-task", "r", "-train", trf, "-test", tef, "-dim", "1,1,8", "-iter", "20", "-method", "sgd",
               "-learn_rate", "0.01", "-regular", "0,0,0.01", "-init_stdev", "0.1", "-seed", "42",
               "-out", os.path.join(td, "pred"), "-save_model", os.path.join(td, "model")]
        r = subprocess.run(cmd, capture_output=True, text=True, check=True)
        iters = [[float(x.split("=")[1]) for x in line.split("\t")[1:3]] for line in r.stdout.splitlines() if line.startswith("#Iter=")]
        pred = np.loadtxt(os.path.join(td, "pred"))
        lines = open(os.path.join(td, "model")).read().splitlines()
        n = 943 + 1682
        w0 = float(lines[1]); w = np.array([float(x) for x in lines[3:3 + n]])
        v = np.array([[float(x) for x in ln.split()] for ln in lines[4 + n:4 + 2 * n]]).T     # file is feature-major
        # the same seed through the harness gives the full-precision initial model (same srand/init order)
        pre = os.path.join(td, "h")
        O.run_ref_harness(["sgd", trf, tef, "r", 1, 1, 8, 0, "0.01", "0", "0", "0.01", "0.1", 42, pre])
        init = O.Model.from_dump(pre + ".init.bin")
    np.savez_compressed(os.path.join(HERE, "c1_ml100k_shaped.npz"),
                        train_entries=tr.entries, train_row_ptr=tr.row_ptr.astype(np.uint32), train_target=tr.target,
                        test_entries=te.entries, test_row_ptr=te.row_ptr.astype(np.uint32), test_target=te.target,
                        stdout_iters=np.array(iters), out_pred=pred, model_w0=w0, model_w=w, model_v=v,
                        init_w0=init.w0, init_w=init.w, init_v=init.v, cmdline=" ".join(cmd[1:]))
    print("c1_ml100k_shaped       stock libFM: last #Iter line Train=%g Test=%g" % tuple(iters[-1]))


def main():
    O.build()
    if "--c1" in sys.argv or not os.path.exists(os.path.join(HERE, "c1_ml100k_shaped.npz")):
        make_c1()
    make_rel()
    make_mcmc()
    make_sgda()
    make_als()
    for name, case in CASES.items():
        gen = getattr(datagen, case["gen"])
        tr = O.Data(*gen(**case["train"]))
        te = O.Data(*gen(**case["test"]))
        cfg = case["cfg"]
        with tempfile.TemporaryDirectory() as td:
            trf, tef, pre = os.path.join(td, "train.libfm"), os.path.join(td, "test.libfm"), os.path.join(td, "out")
            tr.write_libsvm(trf)
            te.write_libsvm(tef)
            O.run_ref_harness(["sgd", trf, tef, cfg["task"], cfg["k0"], cfg["k1"], cfg["k"], cfg["iters"], repr(cfg["lr"]),
                               repr(cfg["reg"][0]), repr(cfg["reg"][1]), repr(cfg["reg"][2]), repr(cfg["init_stdev"]),
                               cfg["seed"], pre])
            init = O.Model.from_dump(pre + ".init.bin")
            final = O.Model.from_dump(pre + ".final.bin")
            pred_raw = np.fromfile(pre + ".pred_raw.bin", dtype=np.float64)
            pred_out = np.fromfile(pre + ".pred_out.bin", dtype=np.float64)
            ev = np.loadtxt(pre + ".eval.txt", ndmin=2)
        # what the reference parser saw: targets as parsed floats; for task c they become +-1 (libfm.cpp:302-306)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            train_entries=tr.entries, train_row_ptr=tr.row_ptr, train_target=tr.target,
            test_entries=te.entries, test_row_ptr=te.row_ptr, test_target=te.target,
            task=cfg["task"], k0=cfg["k0"], k1=cfg["k1"], k=cfg["k"], iters=cfg["iters"], lr=cfg["lr"],
            reg=np.array(cfg["reg"]), init_stdev=cfg["init_stdev"], seed=cfg["seed"],
            n=init.n, init_w0=init.w0, init_w=init.w, init_v=init.v,
            final_w0=final.w0, final_w=final.w, final_v=final.v,
            pred_raw=pred_raw, pred_out=pred_out, eval=ev)
        print("%-22s n=%d k=%d rows=%d/%d  eval[-1]=%s" % (name, init.n, init.k, tr.n_rows, te.n_rows, ev[-1]))


def make_mcmc_seed_band(seeds=range(101, 113), only=None):
    """the REFERENCE's own seed-to-seed distribution on the MCMC fixtures: -seed changes its libc rand() stream (initial
    model AND every draw).  Per fixture: the test metric (RMSE / accuracy) of the posterior-mean prediction for each seed
    and the seed-averaged prediction.  tests/test_gpu_mcmc.py holds the mean and spread of OUR chains to this band."""
    out = {}
    band_file = os.path.join(HERE, "mcmc_ref_seed_band.npz")
    if only:                                                  # add / refresh single fixtures, keep the rest of the band file as it is
        out = dict(np.load(band_file))
    for name, case in MCMC_CASES.items():
        if (only and name not in only) or (not only and name in ("mcmc_reg_ml_k64", "mcmc_reg_ml_k128")):
            continue
        z = np.load(os.path.join(HERE, name + ".npz"))
        cfg = case["cfg"]
        tr = O.Data(z["train_entries"], z["train_row_ptr"], z["train_target"])
        te = O.Data(z["test_entries"], z["test_row_ptr"], z["test_target"])
        y = z["test_target"].astype(np.float64)
        metrics, preds = [], []
        with tempfile.TemporaryDirectory() as td:
            trf, tef = os.path.join(td, "train.libfm"), os.path.join(td, "test.libfm")
            tr.write_libsvm(trf)
            te.write_libsvm(tef)
            env = {}
            if "groups" in case:
                write_meta(os.path.join(td, "meta"), case, case["n_nominal"])
                env = {"FMX_META": os.path.join(td, "meta")}
            for seed in seeds:
                pre = os.path.join(td, "o%d" % seed)
                O.run_ref_harness(["mcmc", trf, tef, cfg["task"], cfg["k0"], cfg["k1"], cfg["k"], cfg["iters"],
                                   repr(cfg["init_stdev"]), seed, pre], env=env)
                p = np.fromfile(pre + ".pred_out.bin", dtype=np.float64)
                preds.append(p)
                metrics.append(np.sqrt(np.mean((p - y) ** 2)) if cfg["task"] == "r" else np.mean((p >= 0.5) == (y > 0)))
        out[name + "_metric"] = np.array(metrics)
        out[name + "_pred_mean"] = np.mean(preds, axis=0)
        # every chain against the mean of the OTHER chains (leave-one-out): the reference's own chain-to-posterior band
        tot = np.sum(preds, axis=0)
        out[name + "_loo_corr"] = np.array([np.corrcoef(p, (tot - p) / (len(preds) - 1))[0, 1] for p in preds])
        out[name + "_loo_rms"] = np.array([np.sqrt(np.mean((p - (tot - p) / (len(preds) - 1)) ** 2)) for p in preds])
        pm = out[name + "_pred_mean"]
        print("%-22s reference over %d seeds: metric mean %.4f sd %.4f (min %.4f max %.4f); one chain vs the seed mean: corr %.4f..%.4f"
              % (name, len(metrics), np.mean(metrics), np.std(metrics, ddof=1), np.min(metrics), np.max(metrics),
                 min(np.corrcoef(p, pm)[0, 1] for p in preds), max(np.corrcoef(p, pm)[0, 1] for p in preds)))
    out["seeds"] = np.array(list(seeds))
    np.savez_compressed(os.path.join(HERE, "mcmc_ref_seed_band.npz"), **out)


if __name__ == "__main__":
    if "--mcmc-only" in sys.argv:    # one MCMC fixture + its 12-seed band, nothing else touched
        names = sys.argv[sys.argv.index("--mcmc-only") + 1].split(",")
        make_mcmc(only=names)
        make_mcmc_seed_band(only=names)
    elif "--mcmc-seed-band" in sys.argv:
        make_mcmc_seed_band()        # does not touch the other fixtures
    else:
        main()
