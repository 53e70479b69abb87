"""rate of FMX_SGD_SEQUENTIAL (the reference's own trajectory on the device) at the bench shape and at a Criteo-shaped one"""
import sys, time
sys.path.insert(0, ".")
from libfm_amd import capi
for name, n, k, nnz, rows, shape in (("uniform n=1e8 k=64 nnz=32", 100_000_000, 64, 32, 20000, capi.SYNTH_UNIFORM),
                                     ("criteo-shaped n=3.3e7 k=64 nnz=39", 33_000_000, 64, 39, 20000, capi.SYNTH_CRITEO),
                                     ("uniform n=1e6 k=8 nnz=16", 1_000_000, 8, 16, 20000, capi.SYNTH_UNIFORM)):
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz, shape)
    h.sgd_epoch(0, capi.SGD_SEQUENTIAL)
    h.synchronize()
    t0 = time.perf_counter()
    st = h.sgd_epoch(0, capi.SGD_SEQUENTIAL)
    h.synchronize()
    dt = time.perf_counter() - t0
    print("%-36s %8.1f k examples/s (%.2f us per example; %d launches-groups, status %#x)" % (name, rows / dt / 1e3, dt / rows * 1e6, st.batches, st.status), flush=True)
    h.close()
