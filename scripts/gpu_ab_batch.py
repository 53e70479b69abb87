"""in-process comparison of batch sizes / bias lags of the one-pass step: the same rows in several slots (segments are per slot and
batch size), epochs round-robin:  python scripts/gpu_ab_batch.py 131072,262144,524288 2,3"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
batches = [int(x) for x in sys.argv[1].split(",")]
lags = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2").split(",")]
h = capi.Handle(100_000_000, 64, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
h.init_params(0.0, 0.01, 1)
cfgs = []
for i, b in enumerate(batches):
    h.synth_rows(i, 123, 0, 1 << 22, 32)
    for lg in lags:
        cfgs.append((i, b, lg))
res = {c: [] for c in cfgs}
for rnd in range(8):
    for c in cfgs:
        st = h.sgd_epoch(c[0], capi.SGD_MINIBATCH, capi.APPLY_FUSED, c[1], 0, capi.FLAG_BIAS_LAG, c[2])
        if rnd >= 2:
            res[c].append(st.device_seconds * 1e3)
for c in cfgs:
    print("batch %7d lag %d: mean %.3f ms  (%.1f M examples/s)" % (c[1], c[2], sum(res[c]) / len(res[c]), (1 << 22) / (sum(res[c]) / len(res[c])) / 1e3))
