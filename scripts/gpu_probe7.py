import os, sys
sys.path.insert(0, ".")
from libfm_amd import capi
rows = 1 << 22
h = capi.Handle(100_000_000, 64, True, True, 1, 0, 0, 0.001, 0.01, -1, 1)
h.init_params(0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, 32)
for _ in range(2):
    h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, 262144, 256)
t = min(h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, 262144, 256).device_seconds for _ in range(5))
print("%s  %7.1f Mex/s" % (os.path.basename(os.environ.get("FMX_LIB", "libfmx.so")), rows / t / 1e6), flush=True)
h.close()
