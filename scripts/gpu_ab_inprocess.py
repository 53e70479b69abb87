"""A/B of two settings of an environment knob that the library reads per epoch, alternating epochs inside ONE process (processes
differ by up to 12 % on the same box):  python scripts/gpu_ab_inprocess.py FMX_FUSED_MERGE 0 1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
knob, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
os.environ[knob] = b                                          # (structures that depend on the knob are built with it set)
h = capi.Handle(100_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
h.init_params(0.0, 0.01, 1)
h.synth_rows(0, 123, 0, 1 << 22, 32)
res = {a: [], b: []}
for i in range(24):
    val = b if (i & 1) else a
    os.environ[knob] = val
    st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 262144, 0, capi.FLAG_BIAS_LAG, 2)
    if i >= 4:
        res[val].append(st.device_seconds * 1e3)
for v in (a, b):
    print("%s=%s: mean %.3f ms  min %.3f  max %.3f" % (knob, v, sum(res[v]) / len(res[v]), min(res[v]), max(res[v])))
