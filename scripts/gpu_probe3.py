"""probe: row layouts (FMX_WPAD floats of padding holding w) for predict / fused / minibatch, n=1e8 k=64 z=32."""
import os, sys
sys.path.insert(0, ".")
from libfm_amd import capi

def probe(n, k, nnz, rows, label):
    h = capi.Handle(n, k, True, True, 1, 0, 0, 0.001, 0.01, -1, 1)
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    h.evaluate(0)
    t = min(h.evaluate(0).device_seconds for _ in range(3))
    out = "%-10s predict %6.1f Mrows/s" % (label, rows / t / 1e6)
    h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, rows, 1024)
    t = min(h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_STORE, rows, 1024, capi.FLAG_TIME_MAIN_KERNEL).main_kernel_seconds for _ in range(3))
    out += " | fused %6.1f Mex/s" % (rows / t / 1e6)
    for ap, nm in ((capi.APPLY_SEGMENTED, "seg"), (capi.APPLY_STORE, "store")):
        h.sgd_epoch(0, capi.SGD_MINIBATCH, ap, 16384, 256)
        st = h.sgd_epoch(0, capi.SGD_MINIBATCH, ap, 16384, 256, capi.FLAG_TIME_MAIN_KERNEL)
        out += " | mb-%s %6.1f Mex/s (apply %5.1f us)" % (nm, rows / st.device_seconds / 1e6, st.main_kernel_seconds / st.main_kernel_launches * 1e6)
    print(out, flush=True)
    h.close()

probe(100_000_000, 64, 32, 1 << 21, "WPAD=" + os.environ.get("FMX_WPAD", "dflt"))
