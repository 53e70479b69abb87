"""compile-time variants of the library (libfm_amd/variants/libfmx_<tag>.so, built with extra -D flags) against the shipped one,
in ONE process: every library gets R handles of its own (a handle's physical placement moves its rate by several per cent, so one
handle per variant would measure the placement), epochs round-robin over all handles.
    python scripts/gpu_ab_variants.py ent,wnt,mw6 [R=2] [what=fused]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
tags = ["base"] + sys.argv[1].split(",")
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
what = sys.argv[3] if len(sys.argv) > 3 else "fused"
base = capi.load()
libs = {"base": base}
for t in tags[1:]:
    L = C.CDLL(os.path.join(os.path.dirname(capi.LIB_PATH), "variants", "libfmx_%s.so" % t))
    for name, res, args in capi.SYMBOLS:
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
    libs[t] = L
hs = []
for r in range(R):
    for t in tags:
        capi._lib = libs[t]
        h = capi.Handle(100_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
        h.init_params(0.0, 0.01, 1)
        h.synth_rows(0, 123, 0, 1 << 22, 32)
        hs.append((t, r, h))
step = {"fused": lambda h: h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 262144, 0, capi.FLAG_BIAS_LAG, 2).device_seconds,
        "predict": lambda h: h.evaluate(0).device_seconds,
        "hogwild": lambda h: h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 262144, 0, 0, 0).device_seconds}[what]
res = {(t, r): [] for t, r, _ in hs}
for rnd in range(8):
    for t, r, h in hs:
        dt = step(h)
        if rnd >= 2:
            res[(t, r)].append(dt * 1e3)
for t in tags:
    per = [sum(res[(t, r)]) / len(res[(t, r)]) for r in range(R)]
    print("%-6s %s: mean %.3f ms   per handle %s" % (t, what, sum(per) / R, " ".join("%.3f" % x for x in per)))
