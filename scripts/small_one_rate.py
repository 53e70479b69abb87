"""small batches (Criteo-shaped rows, BASELINE configs[2]: the library's batch is 512) as two launches per batch vs ONE (FMX_SMALL_ONE=1,
libfm_amd/csrc/fmx_small_kernels.h): examples/s of the default one-pass rule, same handle parameters, same rows"""
import os, sys, time
sys.path.insert(0, ".")
from libfm_amd import capi
n, k, nnz, rows = 33_000_000, 64, 39, 1 << 18
lr = float(os.environ.get("LR", "0.01"))                 # (a smaller step: a larger batch passes the stability cut)
for tag, env in (("two launches per batch", "0"), ("one launch per batch", "1")):
    os.environ["FMX_SMALL_ONE"] = env
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, lr, -1.0, 1.0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz, capi.SYNTH_CRITEO)
    for _ in range(2):
        h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, 2)
    h.synchronize()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, 2)
    h.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-24s %8.2f M examples/s  (%.2f us per batch of %d, %d batches, status %#x)" % (tag, rows / dt / 1e6, dt / st.batches * 1e6, st.batch_used, st.batches, st.status), flush=True)
    h.close()
