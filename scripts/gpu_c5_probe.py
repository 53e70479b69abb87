"""probe: BASELINE.json configs C5 shape -- MCMC (Gibbs draws + per-sweep moments) at n=1e8, k=128 on ONE GPU
(the fp32 table is 51.2 GB: fits the 288 GB of one MI355X, SURVEY section 8 size table)."""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
import numpy as np
from libfm_amd import capi


def probe(n, k, nnz, rows, sweeps=2):
    t0 = time.time()
    h = capi.Handle(n, k, True, True, 1, 0.0, 1.0, 10.0, 0.0, -1.0, 1.0)
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    h.synchronize()
    t_setup = time.time() - t0
    t0 = time.time(); h.als_begin(0); t_begin = time.time() - t0
    out = []
    for it in range(sweeps):
        t0 = time.time(); se2, se, mom = h.als_moments(); t_mom = time.time() - t0
        st = h.als_sweep(1.0 + it, 10.0, 1.0, 0.0, 0.0, True, 7)
        out.append((t_mom, st.device_seconds, st.train_metric, st.levels))
    print("MCMC n=%d k=%d nnz=%d rows=%d: setup %.1fs, begin %.2fs; per sweep: moments %.3f s, draws+repredict %.3f s, levels=%d, train acc %.4f -> %.4f; params %.1f GB"
          % (n, k, nnz, rows, t_setup, t_begin, out[-1][0], out[-1][1], out[-1][3], out[0][2], out[-1][2], h.info().bytes_params / 1e9), flush=True)
    assert np.isfinite(mom).all()
    h.als_end(); h.close()


if __name__ == "__main__":
    probe(100_000_000, 128, 16, 1 << 22)
    probe(100_000_000, 64, 32, 1 << 22)
