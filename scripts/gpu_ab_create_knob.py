"""A/B of an environment knob that the library reads at fmx_create (e.g. FMX_SCAN_CU 0 1): R handles per setting in ONE process
(placement-probed tables), epochs of the default one-pass workload round-robin.  python scripts/gpu_ab_create_knob.py KNOB a b [R=2]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
knob, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
R = int(sys.argv[4]) if len(sys.argv) > 4 else 2
hs = []
for r in range(R):
    for val in (a, b):
        os.environ[knob] = val
        h = capi.Handle(100_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
        h.init_params(0.0, 0.01, 1)
        h.synth_rows(0, 123, 0, 1 << 22, 32)
        hs.append((val, r, h))
res = {(v, r): [] for v, r, _ in hs}
for rnd in range(8):
    for v, r, h in hs:
        st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 262144, 0, capi.FLAG_BIAS_LAG, 2)
        if rnd >= 2:
            res[(v, r)].append(st.device_seconds * 1e3)
for v in (a, b):
    per = [sum(res[(v, r)]) / len(res[(v, r)]) for r in range(R)]
    print("%s=%s: mean %.3f ms   per handle %s" % (knob, v, sum(per) / R, " ".join("%.3f" % x for x in per)))
