"""calibration workload for the PMC byte counters: predict (k_rowsums) with k1=0 over n=1e8, k=64, nnz=32:
every example reads exactly 32 random 256-B rows (no reuse: the table is 25.6 GB) + 256 B of entries + 8 B row_ptr."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from libfm_amd import capi
rows = 1 << 21
h = capi.Handle(100_000_000, 64, True, False, 1, 0, 0, 0.001, 0.01, -1, 1)
h.init_params(0, 0.01, 1)
h.synth_rows(0, 123, 0, rows, 32)
for _ in range(3):
    h.evaluate(0)
h.close()
