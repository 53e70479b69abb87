"""probe: predict rate without linear weights (-dim 1,0,64), north-star shape."""
import sys
sys.path.insert(0, ".")
from libfm_amd import capi
n, k, nnz, rows = 100_000_000, 64, 32, 1 << 22
h = capi.Handle(n, k, True, False, 1, 0, 0, 0.001, 0.01, -1, 1)
h.init_params(0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, nnz)
h.evaluate(0)
t = min(h.evaluate(0).device_seconds for _ in range(4))
print("predict k1=0: %.1f Mrows/s, %.0f GB/s algorithmic read" % (rows / t / 1e6, rows * (nnz * (4 * k + 8) + 4) / t / 1e9))
h.close()
