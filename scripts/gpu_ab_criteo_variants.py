"""compile-time variants on the Criteo-shaped small-batch path (BASELINE configs[2] on one GPU), one process, one handle per library,
epochs round-robin:  python scripts/gpu_ab_criteo_variants.py vsc1"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
tags = ["base"] + sys.argv[1].split(",")
base = capi.load()
libs = {"base": base}
for t in tags[1:]:
    L = C.CDLL(os.path.join(os.path.dirname(capi.LIB_PATH), "variants", "libfmx_%s.so" % t))
    for name, res, args in capi.SYMBOLS:
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
    libs[t] = L
rows = 1 << 20
hs = []
for t in tags:
    capi._lib = libs[t]
    h = capi.Handle(33_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, 39, capi.SYNTH_CRITEO)
    hs.append((t, h))
res = {t: [] for t in tags}
for rnd in range(6):
    for t, h in hs:
        capi._lib = libs[t]
        dt = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, 2).device_seconds
        if rnd >= 2:
            res[t].append(dt * 1e3)
for t in tags:
    ms = sum(res[t]) / len(res[t])
    print("%-6s %.3f ms/epoch  %.2f us/batch  %.1f M ex/s" % (t, ms, ms * 1e3 / (rows / 512), rows / ms / 1e3), flush=True)
