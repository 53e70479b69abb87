"""per-epoch device time of the default bench workload: is there a warm-up drift / run-to-run noise?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
h = capi.Handle(100_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
h.init_params(0.0, 0.01, 1)
h.synth_rows(0, 123, 0, 1 << 22, 32)
if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]))
ts = []
for i in range(40):
    st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 262144, 0, capi.FLAG_BIAS_LAG, 2)
    ts.append(st.device_seconds * 1e3)
print(" ".join("%.2f" % t for t in ts))
