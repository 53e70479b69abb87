"""probe: cost of the separate w[] gather (k1 on/off), predict and fused step, n=1e8 k=64 z=32."""
import sys
sys.path.insert(0, ".")
from libfm_amd import capi

def probe(n, k, nnz, rows, k1, label):
    h = capi.Handle(n, k, True, bool(k1), 1, 0, 0, 0.001, 0.01, -1, 1)
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    h.evaluate(0)
    t = min(h.evaluate(0).device_seconds for _ in range(3))
    print("%-24s predict %7.1f Mrows/s (%.0f GB/s of V rows)" % (label, rows / t / 1e6, rows * nnz * 4 * k / t / 1e9), flush=True)
    h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, rows, 1024)
    t = min(h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, rows, 1024, capi.FLAG_TIME_MAIN_KERNEL).main_kernel_seconds for _ in range(3))
    print("%-24s fused   %7.1f Mex/s   (%.0f GB/s of V rows r+w)" % (label, rows / t / 1e6, 2 * rows * nnz * 4 * k / t / 1e9), flush=True)
    h.close()

rows = 1 << 21
probe(100_000_000, 64, 32, rows, 1, "k1=1")
probe(100_000_000, 64, 32, rows, 0, "k1=0 (no w gather)")
