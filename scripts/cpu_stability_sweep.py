"""CPU (oracle only): where the batch rule leaves the reference's online loop -- the sweep behind the table in DESIGN.md section 3a.
    python scripts/cpu_stability_sweep.py
For every data set: the rows' collision mass C, the batch predicted for gain = lr * curvature * batch * C = 1, and the final test
metric of the online loop and of the batch rule (bias lag 2, micro-chunk 256 / 64) over a ladder of batch sizes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O      # noqa: E402
import datagen as DG                # noqa: E402


def logloss(m, d):
    p = O.predict_raw(m, d)
    with np.errstate(over="ignore", invalid="ignore"):
        return float(np.mean(np.log1p(np.exp(-d.target.astype(np.float64) * p))))


def sweep(name, tr, te, n, k, lr, task, lo, hi, Bs, epochs=4):
    C = DG.collision_mass(tr.entries, tr.n_rows, n)
    curv = 0.25 if task == 1 else 1.0
    print("== %s: n=%d rows=%d C=%.4f  batch at gain 1: %.0f  (gain 2: %.0f)" % (name, n, tr.n_rows, C, 1 / (lr * curv * C), 2 / (lr * curv * C)))
    chunk = 256 if task == 1 else 64

    def run(label, fn):
        m = O.Model(n, k, True, True, 0.0, 0.0, 0.001)
        m.v[:] = O.init_values(1, n, k, 0.01)
        out = []
        for _ in range(epochs):
            fn(m)
            out.append(O.evaluate(m, te, task, lo, hi)[0] if task == 0 else logloss(m, te))
        print("   %-10s" % label, " ".join("%.4f" % o for o in out), flush=True)
    run("online", lambda m: O.sgd_epoch_online(m, tr, task, lr, lo, hi))
    for B in Bs:
        run("B=%d" % B, lambda m: O.sgd_epoch_minibatch(m, tr, task, lr, lo, hi, B, min(chunk, B), bias_lag=2))


if __name__ == "__main__":
    z, N, ntr = 39, 240000, 200000
    e, rp, y, n = DG.criteo_shaped(N, 5, cat_ids=20000)
    tr, te = O.Data(e[:ntr * z], rp[:ntr + 1], y[:ntr]), O.Data(e[ntr * z:], rp[ntr:] - rp[ntr], y[ntr:])
    sweep("criteo-shaped, lr 0.01", tr, te, n, 8, 0.01, 1, -1, 1, (64, 128, 256, 512, 1024, 2048, 4096, 16384))
    sweep("criteo-shaped, lr 0.002", tr, te, n, 8, 0.002, 1, -1, 1, (512, 1024, 2048, 4096, 8192))
    e, rp, y = DG.onehot_fields(16 * 4000, 16, 120000, 3, zipf=0.8)
    sweep("zipf 0.8, 16 fields", O.Data(e[:100000 * 16], rp[:100001], y[:100000]), O.Data(e[100000 * 16:], rp[100000:] - rp[100000], y[100000:]),
          64000, 8, 0.02, 1, -1, 1, (64, 256, 1024, 2048, 4096, 16384))
    e, rp, y = DG.movielens_shaped(943, 1682, 100000, 42)
    sweep("ML-100K-shaped, regression", O.Data(e[:80000 * 2], rp[:80001], y[:80000]), O.Data(e[80000 * 2:], rp[80000:] - rp[80000], y[80000:]),
          2625, 8, 0.01, 0, 1.0, 5.0, (1024, 4096, 16384, 32768, 80000))
