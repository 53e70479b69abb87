"""in-process A/B of the one-pass step at the bench shape: device-side hand-off vs events between the launch stream and the recurrence's
side stream, over batch sizes -- ONE handle (one table placement), one slot per batch size, epochs round-robin.
    python scripts/gpu_ab_handoff.py [262144,131072,65536] [n] [k] [nnz]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libfm_amd import capi
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "262144,131072,65536").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
nnz = int(sys.argv[4]) if len(sys.argv) > 4 else 32
rows = 1 << 22
h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0)
h.init_params(0.0, 0.01, 1)
cfgs = []
for i, b in enumerate(batches):
    h.synth_rows(i, 123, 0, rows, nnz)
    for ev in (0, 1):
        cfgs.append((i, b, ev))
res = {c: [] for c in cfgs}
w0 = {}
for rnd in range(7):
    for c in cfgs:
        st = h.sgd_epoch(c[0], capi.SGD_MINIBATCH, capi.APPLY_FUSED, c[1], 0, capi.FLAG_EVENT_SYNC if c[2] else 0, 2)
        if rnd >= 2:
            res[c].append(st.device_seconds * 1e3)
for c in cfgs:
    ms = sum(res[c]) / len(res[c])
    nb = (rows + c[1] - 1) // c[1]
    per = 16900 if (k, nnz) == (64, 32) else nnz * (8 * k + 16) + 4
    print("batch %7d %-8s: %.3f ms/epoch  %.4f ms/batch  %.1f M ex/s  frac %.4f" % (c[1], "events" if c[2] else "hand-off", ms, ms / nb, rows / ms / 1e3,
                                                                              rows * per / (ms * 1e-3) / 8e12), flush=True)
