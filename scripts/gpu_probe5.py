"""probe: inter-launch gaps of the hogwild epoch: with / without the bias recurrence (k0), one / two launch streams."""
import os, sys
sys.path.insert(0, ".")
from libfm_amd import capi
rows = 1 << 22
for k0 in (True, False):
    h = capi.Handle(100_000_000, 64, k0, True, 1, 0, 0, 0.001, 0.01, -1, 1)
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, 32)
    for M in (262144, 524288):
        h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, M, 256)
        t = min(h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, M, 256).device_seconds for _ in range(4))
        print("%s k0=%d M=%-7d %7.1f Mex/s  %.3f ms per launch" % (os.environ.get("FMX_HOGWILD_TWO_STREAMS", "one"), k0, M, rows / t / 1e6, t / (rows / M) * 1e3), flush=True)
    h.close()
