"""BASELINE.json configs[0]: "MovieLens-100K in libsvm format, k=8, SGD 20 iters, single-thread CPU via reference
libFM binary (plumbing)".  MovieLens itself is not on disk (no network), so the fixture is an ML-100K-SHAPED set
(943 users + 1682 items, 80 000 / 20 000 rows, integer ratings 1..5 from a planted model) run through the STOCK
reference binary (tests/golden/make_golden.py --c1): its stdout #Iter lines, -out file and -save_model file are
the expected values (6 significant digits, the reference's ostream precision).

CPU part: the pinned oracle reproduces the stock binary's printed numbers.
GPU part: the SEQUENTIAL mode (reference trajectory) driven through the host mirror reproduces them too."""
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

Z = np.load(os.path.join(GOLDEN_DIR, "c1_ml100k_shaped.npz"))
N, K, ITERS, LR, REGV = 943 + 1682, 8, 20, 0.01, 0.01


def sixdigits(a, b):
    """b was printed with 6 significant digits"""
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-8)


def test_oracle_reproduces_the_stock_binary(oracle):
    O = oracle
    tr = O.Data(Z["train_entries"], Z["train_row_ptr"].astype(np.uint64), Z["train_target"])
    te = O.Data(Z["test_entries"], Z["test_row_ptr"].astype(np.uint64), Z["test_target"])
    m = O.Model(N, K, True, True, 0.0, 0.0, REGV)
    m.w0, m.w[:], m.v[:] = float(Z["init_w0"]), Z["init_w"], Z["init_v"]
    lo, hi = float(tr.target.min()), float(tr.target.max())
    lines = []
    for _ in range(ITERS):
        O.sgd_epoch_online(m, tr, 0, LR, lo, hi)
        lines.append([O.evaluate(m, tr, 0, lo, hi)[0], O.evaluate(m, te, 0, lo, hi)[0]])
    sixdigits(np.array(lines), Z["stdout_iters"])
    sixdigits(O.predict_out(m, te, 0, lo, hi), Z["out_pred"])
    sixdigits(m.w0, float(Z["model_w0"]))
    np.testing.assert_allclose(m.w, Z["model_w"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(m.v, Z["model_v"], rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
def test_gpu_sequential_reproduces_the_stock_binary():
    from libfm_amd import learner as L
    train = L.Data(Z["train_entries"], Z["train_row_ptr"].astype(np.uint64), Z["train_target"])
    test = L.Data(Z["test_entries"], Z["test_row_ptr"].astype(np.uint64), Z["test_target"])
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.regv = N, K, REGV
    fm.w0, fm.w, fm.v = float(Z["init_w0"]), Z["init_w"].copy(), Z["init_v"].copy()
    l = L.FMLearnSGD()
    l.fm, l.task, l.num_iter, l.learn_rate, l.mode = fm, 0, ITERS, LR, "sequential"
    l.min_target, l.max_target = train.min_target, train.max_target
    l.out = io.StringIO()
    l.init()
    l.learn(train, test)
    lines = [[float(x.split("=")[1]) for x in ln.split("\t")[1:3]] for ln in l.out.getvalue().splitlines() if ln.startswith("#Iter=")]
    np.testing.assert_allclose(np.array(lines), Z["stdout_iters"], rtol=1e-4)
    np.testing.assert_allclose(l.predict(test), Z["out_pred"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(fm.v, Z["model_v"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(fm.w, Z["model_w"], rtol=1e-4, atol=2e-5)
    l.close()
