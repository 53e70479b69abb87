"""ad-hoc GPU probes: predict-only gather rate and fused-step rate vs table size / occupancy."""
import sys, time, json
sys.path.insert(0, ".")
from libfm_amd import capi

def probe(n, k, nnz, rows, label):
    h = capi.Handle(n, k, True, True, 1, 0, 0, 0.001, 0.01, -1, 1)
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    h.evaluate(0)
    ev = [h.evaluate(0).device_seconds for _ in range(3)]
    t = min(ev)
    rd = nnz * (4 * k + 12) + 4
    print("%-28s predict  %7.1f Mrows/s  %7.1f GB/s algorithmic-read" % (label, rows / t / 1e6, rows * rd / t / 1e9), flush=True)
    for ap, name in ((capi.APPLY_STORE, "store"),):
        h.sgd_epoch(0, capi.SGD_HOGWILD, ap, rows, 1024)
        st = [h.sgd_epoch(0, capi.SGD_HOGWILD, ap, rows, 1024, capi.FLAG_TIME_MAIN_KERNEL) for _ in range(3)]
        t = min(s.main_kernel_seconds for s in st)
        tot = rd + nnz * (4 * k + 4)
        print("%-28s fused-%s %7.1f Mex/s   %7.1f GB/s algorithmic r+w (device %0.2f ms)" % (label, name, rows / t / 1e6, rows * tot / t / 1e9, min(s.device_seconds for s in st) * 1e3), flush=True)
    h.close()

rows = 1 << 21
probe(100_000_000, 64, 32, rows, "n=1e8 k=64 z=32")
probe(10_000_000, 64, 32, rows, "n=1e7 k=64 z=32")
probe(1_000_000, 64, 32, rows, "n=1e6 k=64 z=32")
probe(100_000_000, 128, 32, rows // 2, "n=1e8 k=128 z=32")
probe(10_000_000, 32, 16, rows, "n=1e7 k=32 z=16 (C2)")
probe(33_000_000, 64, 39, rows, "n=3.3e7 k=64 z=39 (C3 shape)")
