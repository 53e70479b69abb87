"""repeat one form of tests/test_gpu_fuzz.py's case `seed` many times and count the runs whose bias differs from the first (a race shows as a rare deviation)"""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import test_gpu_fuzz as F
from libfm_amd import capi
from oracle import oracle as O
seed, reps = int(sys.argv[1]), int(sys.argv[2])
form = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n, k, task, (ent, rp, y), batch, chunk, lag, k0, k1 = F._case(seed)
lo, hi = (float(y.min()), float(y.max())) if task == 0 else (-1.0, 1.0)
lr = 0.003
import inspect
src = inspect.getsource(F.test_random_shape_every_form_of_the_rule)
for ln in src.splitlines():
    if "lr =" in ln and "lr" in ln.split("=")[0]:
        print("test's own:", ln.strip())
forms = [(capi.APPLY_FUSED, 0, lag), (capi.APPLY_DEFAULT, capi.FLAG_BIAS_LAG, lag), (capi.APPLY_SEGMENTED, capi.FLAG_BIAS_LAG, lag), (capi.APPLY_DEFAULT, 0, 0)]
apply_, flags, lg = forms[form]
print("seed", seed, "n", n, "k", k, "task", task, "batch", batch, "chunk", chunk, "lag", lag, "rows", len(y), "form", forms[form], flush=True)
vals = []
for i in range(reps):
    h = capi.Handle(n, k, k0, k1, task, 0.001 if k0 else 0.0, 0.002, 0.004, lr, lo, hi)
    h.set_params(0.02 if k0 else 0.0, O.init_values(22 + seed, n, 1, 0.05)[0] if k1 else np.zeros(n), O.init_values(21 + seed, n, k, 0.05))
    h.upload_rows(0, ent, rp, y)
    for _ in range(2):
        h.sgd_epoch(0, capi.SGD_MINIBATCH, apply_, batch, chunk, flags, lg)
    w0, w, v = h.get_params()
    vals.append((w0, float(np.abs(v).sum())))
    h.close()
base = vals[0]
bad = [i for i, x in enumerate(vals) if abs(x[0] - base[0]) > 1e-7 or abs(x[1] - base[1]) > 1e-4 * abs(base[1])]
print("runs", reps, "deviating from the first:", len(bad), bad[:10], "first", base, "e.g.", [vals[i] for i in bad[:3]])
