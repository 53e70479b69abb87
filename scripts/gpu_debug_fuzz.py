"""debug: the large-batch fuzz case of a seed under every flag combination -- which of {bias, w, v, predictions with / without the side stream} leave the oracle"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from libfm_amd import capi
from oracle import oracle
for seed in [int(x) for x in sys.argv[1].split(",")]:
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.choice([4, 8, 16, 33, 64])); task = int(rng.integers(0, 2)); rows = int(rng.integers(66000, 120000))
    if rng.integers(0, 2):
        nnz = int(rng.choice([4, 9, 16])); n = nnz * int(rng.integers(2000, 40000))
        ent, rp, y = datagen.onehot_fields(n, nnz, rows, seed, zipf=float(rng.choice([0.0, 0.9])), classification=bool(task)); kind = "fields"
    else:
        n = int(rng.integers(20000, 200000))
        ent, rp, y = datagen.ragged_real(n, rows, int(rng.integers(3, 20)), seed, classification=bool(task), empty_every=int(rng.choice([0, 13]))); kind = "ragged"
    batch = int(rng.choice([32768, 33001, 50000, 65536])); chunk = int(rng.choice([64, 256, 300, 512])); lag = int(rng.integers(1, 5))
    flags0 = (capi.FLAG_EVENT_SYNC if rng.integers(0, 2) else 0) | (capi.FLAG_KEEP_WSIDE if rng.integers(0, 2) else 0)
    lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
    lr = min(0.004, 0.9 / (chunk * (1.0 if task == 0 else 0.25)))
    d = oracle.Data(ent, rp, y)
    print("seed %d: %s n=%d k=%d task=%d rows=%d nnz=%d batch=%d chunk=%d lag=%d flags0=%d lr=%g y in [%g,%g]" % (seed, kind, n, k, task, rows, len(ent), batch, chunk, lag, flags0, lr, lo, hi), flush=True)
    m = oracle.Model(n, k, True, True, 0.001, 0.002, 0.004)
    m.v[:] = oracle.init_values(31 + seed, n, k, 0.05); m.w[:] = oracle.init_values(32 + seed, n, 1, 0.05)[0]; m.w0 = 0.02
    for ep in range(2):
        oracle.sgd_epoch_minibatch(m, d, task, lr, lo, hi, batch, chunk, bias_lag=lag)
    o_p = oracle.predict_raw(m, d)
    for flags in (0, 16, 32, 48):
        for apply_ in (capi.APPLY_FUSED, capi.APPLY_SEGMENTED):
            if apply_ == capi.APPLY_SEGMENTED and flags:
                continue
            h = capi.Handle(n, k, True, True, task, 0.001, 0.002, 0.004, lr, lo, hi)
            h.set_params(0.02, oracle.init_values(32 + seed, n, 1, 0.05)[0], oracle.init_values(31 + seed, n, k, 0.05))
            h.upload_rows(0, ent, rp, y)
            for ep in range(2):
                h.sgd_epoch(0, capi.SGD_MINIBATCH, apply_, batch, chunk, flags | (capi.FLAG_BIAS_LAG if apply_ == capi.APPLY_SEGMENTED else 0), lag)
            ev = h.evaluate(0)
            p1 = h.predict(0, d.n_rows)
            w0, w, v = h.get_params()
            h.set_params(w0, w, v)                                   # (makes any side stream stale: the next pass gathers)
            p2 = h.predict(0, d.n_rows)
            bad = np.abs(p1 - o_p) > 1e-4 * np.abs(o_p) + 1e-3
            print("  apply %d flags %2d wside=%d: |w0| %.2e  w max %.2e (%d > 1e-4)  v max %.2e (%d > 1e-4)  pred(stream) max %.2e bad rows %d  pred(gather) max %.2e  stream-vs-gather max %.2e"
                  % (apply_, flags, ev.flags & 1, abs(w0 - m.w0), np.abs(w - m.w).max(), int((np.abs(w - m.w) > 1e-4 * np.abs(m.w) + 1e-4).sum()),
                     np.abs(v - m.v).max(), int((np.abs(v - m.v) > 1e-4 * np.abs(m.v) + 1e-4).sum()), np.abs(p1 - o_p).max(), int(bad.sum()),
                     np.abs(p2 - o_p).max(), np.abs(p1 - p2).max()), flush=True)
            if bad.any():
                r = np.flatnonzero(bad)[:5]
                print("    first bad rows", r, "batch index", r // batch, "row sizes", (rp[r + 1] - rp[r]))
            h.close()
