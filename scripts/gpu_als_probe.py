"""probe: ALS iteration time on field-structured synthetic data (config C4 shape: n=1e7, k=64, 16 nnz/row)."""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
import numpy as np
from libfm_amd import capi

def probe(n, k, nnz, rows, sweeps=2):
    h = capi.Handle(n, k, True, True, 0, 0.0, 1.0, 10.0, 0.0, -1.0, 1.0)
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    t0 = time.time(); h.als_begin(0); t_begin = time.time() - t0
    st = [h.als_sweep(1.0, 10.0) for _ in range(sweeps)]
    print("ALS n=%d k=%d nnz=%d rows=%d: begin %.2fs (segments+levels), sweep %.3f s, levels=%d, train rmse %.4f -> %.4f"
          % (n, k, nnz, rows, t_begin, st[-1].device_seconds, st[-1].levels, st[0].train_metric, st[-1].train_metric), flush=True)
    h.als_end(); h.close()

if __name__ == "__main__":
    probe(10_000_000, 64, 16, 1 << 22)
