#!/usr/bin/env python3
"""The arithmetic of k_scan_pit (libfm_amd/csrc/fmx_kernels.h) on the CPU: the bias recurrence of one batch

    w_{c+1} = w_c - lr (sum_{i in chunk c} m_i(w_c) + n_c reg0 w_c)          fm_sgd.h:34-37 summed per micro-chunk,
                                                                              m_i = the multiplier of fm_learn_sgd_element.h:58-65

solved by Newton's method on the whole path (every chunk's step linearised around the previous iterate's path, the linearised chain -- an
affine recurrence -- solved as a prefix "scan"), in fp32 like the device, against the serial chain in fp64.  Prints the path change per
iteration and the distance of the two end values.  tests/test_pit_arithmetic.py asserts it.

    python scripts/cpu_pit_check.py [--rows 262144] [--chunk 32] [--lr 0.01] [--task 1]"""
import argparse
import json

import numpy as np

TOL, MAX_IT = 5e-4, 12          # PIT_TOL, PIT_MAX_IT


def multiplier(task, p, y, lo, hi):
    """(m, dm/dp): fm_learn_sgd_element.h:60-65"""
    if task == 1:
        inv = 1.0 / (1.0 + np.exp(y * p))
        return -y * inv, (y * y) * inv * (1.0 - inv)
    pc = np.clip(p, lo, hi)
    return pc - y, ((p > lo) & (p < hi)).astype(p.dtype)


def serial(rest, y, chunk, lr, reg0, w0, task, lo=-1.0, hi=1.0):
    w = np.float64(w0)
    for c0 in range(0, len(rest), chunk):
        r, yy = rest[c0:c0 + chunk].astype(np.float64), y[c0:c0 + chunk].astype(np.float64)
        m, _ = multiplier(task, np.float64(np.float32(w)) + r, yy, lo, hi)      # (the device rounds the bias to fp32 per chunk)
        w = w - lr * (m.sum() + len(r) * reg0 * np.float64(np.float32(w)))
    return float(w)


def newton(rest, y, chunk, lr, reg0, w0, task, lo=-1.0, hi=1.0, ft=np.float32):
    n = len(rest)
    nc = (n + chunk - 1) // chunk
    pad = nc * chunk - n
    r = np.concatenate([rest.astype(ft), np.zeros(pad, ft)])
    yy = np.concatenate([y.astype(ft), np.zeros(pad, ft)])
    ok = np.arange(nc * chunk) < n
    cnt = np.minimum(chunk, n - np.arange(nc) * chunk).astype(ft)
    w0s = ft(w0)
    d = np.zeros(nc + 1, ft)
    changes = []
    for _ in range(MAX_IT):
        p = (w0s + np.repeat(d[:nc], chunk)) + r
        m, dm = multiplier(task, p, yy, ft(lo), ft(hi))
        m, dm = np.where(ok, m, 0).astype(ft), np.where(ok, dm, 0).astype(ft)
        F = m.reshape(nc, chunk).sum(1, dtype=ft) + cnt * ft(reg0) * (w0s + d[:nc])
        D = dm.reshape(nc, chunk).sum(1, dtype=ft) + cnt * ft(reg0)
        A, B = (1 - ft(lr) * D).astype(ft), (-ft(lr) * (F - D * d[:nc])).astype(ft)
        nd = np.zeros(nc + 1, ft)
        acc = ft(0)
        for c in range(nc):                        # (the device composes these maps as a parallel prefix scan: same result up to rounding)
            acc = ft(A[c] * acc + B[c])
            nd[c + 1] = acc
        changes.append(float(np.abs(nd - d).max()))
        d = nd
        if changes[-1] < TOL:
            break
    return float(np.float64(w0) + np.float64(d[nc])), changes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144)
    ap.add_argument("--chunk", type=int, default=32)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--task", type=int, default=1)
    ap.add_argument("--skew", type=float, default=0.5, help="share of +1 targets (0.5: the bias hardly drifts)")
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    rest = (rng.standard_normal(a.rows) * 0.2).astype(np.float32)
    y = np.where(rng.random(a.rows) < a.skew, 1.0, -1.0).astype(np.float32)
    ws = serial(rest, y, a.chunk, a.lr, 0.0, 0.0, a.task)
    wn, ch = newton(rest, y, a.chunk, a.lr, 0.0, 0.0, a.task)
    print(json.dumps({"rows": a.rows, "chunk": a.chunk, "serial": ws, "newton": wn, "abs_diff": abs(ws - wn), "path_change_per_iteration": ch}))


if __name__ == "__main__":
    main()
