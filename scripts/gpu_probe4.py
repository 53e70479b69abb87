"""probe: hogwild epoch throughput vs macro-batch size and stream overlap."""
import os, sys
sys.path.insert(0, ".")
from libfm_amd import capi
rows = 1 << 22
h = capi.Handle(100_000_000, 64, True, True, 1, 0, 0, 0.001, 0.01, -1, 1)
h.init_params(0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, 32)
for M in (65536, 262144, 1048576, 4194304):
    h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, M, 256)
    st = [h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, M, 256) for _ in range(3)]
    t = min(s.device_seconds for s in st)
    print("%s M=%-8d %7.1f Mex/s (epoch %.2f ms)" % (os.environ.get("FMX_HOGWILD_ONE_STREAM", "two"), M, rows / t / 1e6, t * 1e3), flush=True)
h.close()
