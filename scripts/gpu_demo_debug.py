import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from libfm_amd import capi
def run(chunk, rows=200000):
    h = capi.Handle(640000, 16, True, True, 1, 0, 0, 0.001, 0.02, -1, 1)
    h.init_params(0.0, 0.05, 1)
    h.synth_rows(0, 123, 0, rows, 16)
    for it in range(3):
        st = h.sgd_epoch(0, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 0, chunk)
    ev = h.evaluate(0)
    print("chunk", chunk, "rows", rows, "one_wave", os.environ.get("FMX_SCAN_ONE_WAVE"), "acc %.4f w0 %.5f" % (ev.accuracy, h.get_w0()), flush=True)
    h.close()
for c in (0, 256, 1024, 2048):
    run(c)
run(0, 262144); run(0, 300000)
