"""probe: does the overlapped bias recurrence limit the hogwild epoch?  Same epoch with and without the bias (k0)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from libfm_amd import capi
n, k, nnz, rows = 100_000_000, 64, 32, 1 << 22
for k0 in (True, False, True, False):
    h = capi.Handle(n, k, k0, True, 1, 0, 0, 0.001, 0.01, -1, 1)
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    for _ in range(2):
        h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 0, 0)
    t = min(h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 0, 0).device_seconds for _ in range(5))
    print("k0=%s: hogwild %.1f Mex/s" % (k0, rows / t / 1e6), flush=True)
    h.close()
