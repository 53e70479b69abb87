#!/usr/bin/env python3
"""How far does the headline batch rule end from the reference's ONLINE loop at the bench shape?  (CPU, oracle only.)

The product's headline mode is the restated minibatch rule (oracle fmo_sgd_epoch_minibatch_ex: batch 262 144, micro-chunk 256,
bias lag 2); the reference is strictly online (fm_learn_sgd_element.h:56-67 calling fm_sgd.h:33-51 per row).  Both loops live in
the oracle, so the distance is one CPU-side comparison on the SUB-MODEL of the rows' features: n = 1e8 does not fit the host, but
untouched parameters do not enter the arithmetic.  Start values are fmx_init_params' (fmo_init_value keyed by the GLOBAL id).

    python scripts/cpu_online_vs_rule.py [--rows 278528] [--n 100000000] [--k 64] [--nnz 32] [--batch 262144] [--lag 2]

Prints one JSON line: deviations of the bias, of the touched parameter rows and of the predictions for the epoch's own rows.
tests/test_gpu_fullsize.py asserts the band stated in DESIGN.md section 3 with the DEVICE in place of the oracle's rule."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def deviation(O, m_rule, m_online, d):
    """the figures of the band: predictions of the epoch's rows under both models, parameters of the touched rows"""
    p_rule, p_on = O.predict_raw(m_rule, d), O.predict_raw(m_online, d)
    rms = float(np.sqrt(np.mean(p_on ** 2)))
    dp = np.abs(p_rule - p_on)
    dv = np.abs(m_rule.v - m_online.v)
    vmax = float(np.abs(m_online.v).max())
    return {"pred_rms": round(rms, 6), "pred_max_abs": float(dp.max()), "pred_mean_abs": float(dp.mean()),
            "pred_max_rel_to_rms": float(dp.max() / rms), "pred_mean_rel_to_rms": float(dp.mean() / rms),
            # the bias is the one parameter every row shares: most of the prediction distance is the distance of the two bias paths
            "w0_rule": float(m_rule.w0), "w0_online": float(m_online.w0), "w0_abs": float(abs(m_rule.w0 - m_online.w0)),
            "pred_max_rel_to_rms_without_bias": float(np.abs((p_rule - m_rule.w0) - (p_on - m_online.w0)).max() / rms),
            "v_max_abs": float(dv.max()), "v_max_rel_to_vmax": float(dv.max() / vmax), "v_mean_abs": float(dv.mean()),
            "w_max_abs": float(np.abs(m_rule.w - m_online.w).max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144 + 16384)
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--nnz", type=int, default=32)
    ap.add_argument("--batch", type=int, default=262144)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--lag", type=int, default=2)
    ap.add_argument("--stdev", type=float, default=0.01, help="fmx_init_params stdev (bench.py: 0.01)")
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--regv", type=float, default=0.001)
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--order-noise", action="store_true",
                    help="instead of the batch rule: the reference's ONLINE loop over the same rows with every two neighbours swapped -- how far the "
                         "reference's own result moves under the smallest change of row order (the yardstick for the rule's distance)")
    ap.add_argument("--order-window", type=int, default=0,
                    help="with --order-noise: instead of swapping neighbours, shuffle the rows randomly inside windows of this many rows "
                         "(262144: what the ORDER of the rows inside one batch is worth to the reference itself)")
    a = ap.parse_args()
    from oracle import oracle as O
    t0 = time.time()
    d = O.synth_rows(a.seed, 0, a.rows, a.nnz, a.n)
    ids = np.unique(d.entries["id"])
    ent = d.entries.copy()
    ent["id"] = np.searchsorted(ids, d.entries["id"]).astype(np.uint32)
    ds = O.Data(ent, d.row_ptr, d.target)
    m0 = O.Model(len(ids), a.k, True, True, 0.0, 0.0, a.regv)
    m0.v[:] = O.init_values_ids(1, ids, a.k, a.stdev).astype(np.float32)       # (the device holds fp32)
    m_rule, m_on = m0, m0.copy()          # (two fp64 sub-models: 1.18 M rows touch 37 M features = 19 GB each)
    ds_swapped = None
    if a.order_noise:                     # rows 2i and 2i + 1 change places (fixed row length: the entries move as blocks)
        nrow = a.rows - a.rows % 2
        perm = np.arange(a.rows)
        perm[0:nrow:2], perm[1:nrow:2] = np.arange(1, nrow, 2), np.arange(0, nrow, 2)
        if a.order_window > 1:
            rng = np.random.default_rng(7)
            perm = np.concatenate([w0 + rng.permutation(min(a.order_window, a.rows - w0)) for w0 in range(0, a.rows, a.order_window)])
        ent2 = ent.reshape(a.rows, a.nnz)[perm].reshape(-1).copy()
        ds_swapped = O.Data(ent2, d.row_ptr, d.target[perm].copy())
    for _ in range(a.epochs):
        O.sgd_epoch_online(m_on, ds, 1, a.lr, -1.0, 1.0)
        if a.order_noise:
            O.sgd_epoch_online(m_rule, ds_swapped, 1, a.lr, -1.0, 1.0)
        else:
            O.sgd_epoch_minibatch(m_rule, ds, 1, a.lr, -1.0, 1.0, a.batch, a.chunk, bias_lag=a.lag)
    out = {"what": ("online loop vs online loop over rows shuffled inside windows of %d" % a.order_window if a.order_window > 1 else "online loop vs online loop over pair-swapped rows") if a.order_noise else "batch rule vs online loop", "rows": a.rows, "n": a.n, "k": a.k, "nnz": a.nnz, "batch": a.batch, "chunk": a.chunk, "bias_lag": a.lag, "epochs": a.epochs,
           "stdev": a.stdev, "touched_features": int(len(ids)), "seconds": None}
    out.update(deviation(O, m_rule, m_on, ds))
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
