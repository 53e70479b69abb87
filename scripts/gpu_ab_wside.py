"""what the weight side stream costs the epoch that keeps it and buys the pass that reads it, at the bench shape -- ONE handle, epochs
and evaluation passes alternating:  python scripts/gpu_ab_wside.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
rows = 1 << 22
h = capi.Handle(100_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
h.init_params(0.0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, 32)
res = {"epoch plain": [], "epoch keep": [], "eval gather": [], "eval stream": []}
for rnd in range(8):
    a = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, 2).device_seconds
    ev = h.evaluate(0); assert not (ev.flags & capi.EVAL_WSIDE)
    b = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, capi.FLAG_KEEP_WSIDE, 2).device_seconds
    ev2 = h.evaluate(0); assert ev2.flags & capi.EVAL_WSIDE
    if rnd >= 2:
        res["epoch plain"].append(a); res["epoch keep"].append(b); res["eval gather"].append(ev.device_seconds); res["eval stream"].append(ev2.device_seconds)
for k, v in res.items():
    ms = sum(v) / len(v) * 1e3
    print("%-12s %.3f ms  %.1f M rows/s" % (k, ms, rows / ms / 1e3), flush=True)
