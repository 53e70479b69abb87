"""probe: per-kernel-variant A/B inside one process (same box): fused kernel launch geometries."""
import os, sys
sys.path.insert(0, ".")
from libfm_amd import capi
rows = 1 << 22
h = capi.Handle(100_000_000, 64, True, True, 1, 0, 0, 0.001, 0.01, -1, 1)
h.init_params(0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, 32)
def run(M, reps=4):
    h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, M, 256)
    return min(h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, M, 256).device_seconds for _ in range(reps))
for over in ("1", "2", "3", "4", "6"):
    os.environ["FMX_GRID_OVER"] = over
    t = run(262144)
    print("over=%s M=262144 %7.1f Mex/s" % (over, rows / t / 1e6), flush=True)
h.close()
