"""development aid (not collected): BASELINE configs[0] end to end -- stock libFM binary (CPU) vs python -m libfm_amd.cli
(GPU) on the ML-100K-shaped fixture, same flags; prints wall-clock of the learn phase and the final test RMSE."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402

z = np.load(os.path.join(HERE, "golden", "c1_ml100k_shaped.npz"))
with tempfile.TemporaryDirectory() as td:
    trf, tef = os.path.join(td, "tr"), os.path.join(td, "te")
    O.Data(z["train_entries"], z["train_row_ptr"].astype(np.uint64), z["train_target"]).write_libsvm(trf)
    O.Data(z["test_entries"], z["test_row_ptr"].astype(np.uint64), z["test_target"]).write_libsvm(tef)
    common = ["-task", "r", "-train", trf, "-test", tef, "-dim", "1,1,8", "-iter", "20", "-method", "sgd",
              "-learn_rate", "0.01", "-regular", "0,0,0.01", "-init_stdev", "0.1", "-seed", "42"]
    if os.path.exists(O.REF_LIBFM):
        t = time.time()
        r = subprocess.run([O.REF_LIBFM] + common, capture_output=True, text=True)
        print("stock libFM (1 CPU thread): %.2f s total; %s" % (time.time() - t, [l for l in r.stdout.splitlines() if l.startswith("#Iter")][-1]))
    for mode in ("sequential", "minibatch", "hogwild"):
        t = time.time()
        r = subprocess.run([sys.executable, "-m", "libfm_amd.cli"] + common + ["-gpu_mode", mode], capture_output=True, text=True,
                           cwd=os.path.dirname(HERE))
        last = [l for l in r.stdout.splitlines() if l.startswith("#Iter")]
        print("libfm_amd.cli -gpu_mode %-10s: %.2f s total (python + torch import included); %s" % (mode, time.time() - t, last[-1] if last else r.stderr[-300:]))
