"""A/B of two settings of an environment knob that the library reads per launch / per epoch, alternating steps inside ONE process
(processes differ by up to 12 % on the same box):
    python scripts/gpu_ab_inprocess.py FMX_GRID_OVER 2 64 [fused|twopass|hogwild|predict|als]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
knob, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
what = sys.argv[4] if len(sys.argv) > 4 else "fused"
os.environ[knob] = b                                          # (structures that depend on the knob are built with it set)
if what == "als":
    h = capi.Handle(10_000_000, 64, True, True, capi.TASK_REGRESSION, 0.0, 1.0, 10.0, 0.0, -1.0, 1.0, device=0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, 1 << 22, 16)
    h.als_begin(0)
    step = lambda: h.als_sweep(1.0, 10.0).device_seconds
else:
    h = capi.Handle(100_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, 1 << 22, 32)
    step = {"fused": lambda: h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 262144, 0, capi.FLAG_BIAS_LAG, 2).device_seconds,
            "twopass": lambda: h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 131072, 0, capi.FLAG_BIAS_LAG, 1).device_seconds,
            "hogwild": lambda: h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 262144, 0, 0, 0).device_seconds,
            "predict": lambda: h.evaluate(0).device_seconds}[what]
res = {a: [], b: []}
n = 12 if what == "als" else 24
for i in range(n):
    val = b if (i & 1) else a
    os.environ[knob] = val
    t = step()
    if i >= 4:
        res[val].append(t * 1e3)
for v in (a, b):
    print("%s %s=%s: mean %.3f ms  min %.3f  max %.3f" % (what, knob, v, sum(res[v]) / len(res[v]), min(res[v]), max(res[v])))
