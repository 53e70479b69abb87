"""PMC experiment: predict with k1=1 (w gather on) for different w-load cache policies (FMX_LIB selects the build)."""
import sys
sys.path.insert(0, "/root/repo")
from libfm_amd import capi
rows = 1 << 21
h = capi.Handle(100_000_000, 64, True, True, 1, 0, 0, 0.001, 0.01, -1, 1)
h.init_params(0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, 32)
for _ in range(3):
    ev = h.evaluate(0)
print("predict %.1f Mrows/s" % (rows / ev.device_seconds / 1e6))
h.close()
