"""variant probe: predict / hogwild / minibatch rates for one libfmx build (FMX_LIB), north-star shape."""
import sys, os
sys.path.insert(0, ".")
from libfm_amd import capi

n, k, nnz, rows = 100_000_000, 64, 32, 1 << 22
if len(sys.argv) > 1:
    n, k, nnz = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
h = capi.Handle(n, k, True, True, 1, 0, 0, 0.001, 0.01, -1, 1)
h.init_params(0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, nnz)
h.evaluate(0)
tp = min(h.evaluate(0).device_seconds for _ in range(4))
h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 0, 256)
th = min(h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 0, 256).device_seconds for _ in range(4))
h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 16384, 256, capi.FLAG_BIAS_LAG)
tm = min(h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 16384, 256, capi.FLAG_BIAS_LAG).device_seconds for _ in range(3))
h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 131072, 256, capi.FLAG_BIAS_LAG)
tM = min(h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 131072, 256, capi.FLAG_BIAS_LAG).device_seconds for _ in range(3))
print("%-16s n=%.0e k=%d z=%d: predict %6.1f Mrows/s | hogwild %6.1f | minibatch B=16k %6.1f  B=128k %6.1f Mex/s"
      % (os.path.basename(os.environ.get("FMX_LIB", "libfmx.so")), n, k, nnz, rows / tp / 1e6, rows / th / 1e6, rows / tm / 1e6, rows / tM / 1e6), flush=True)
h.close()
