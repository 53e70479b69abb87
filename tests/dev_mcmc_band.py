"""development aid (not collected by pytest): the spread of OUR sampler against the reference fixture over several
seeds, for calibrating the statistical bars of test_gpu_mcmc.py / test_gpu_adapter.py.  Needs a GPU."""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from common import Golden  # noqa: E402
from oracle import oracle as O  # noqa: E402

HARNESS = os.path.join(O.REF_DIR, "ref_harness_gpu")

for name in sys.argv[1:] or ["mcmc_reg_ml_groups", "mcmc_reg_ml"]:
    g = Golden(name)
    z = g.z
    ref = z["pred_out"]
    preds = []
    with tempfile.TemporaryDirectory() as td:
        trf, tef = os.path.join(td, "tr"), os.path.join(td, "te")
        O.Data(z["train_entries"], z["train_row_ptr"], z["train_target"]).write_libsvm(trf)
        O.Data(z["test_entries"], z["test_row_ptr"], z["test_target"]).write_libsvm(tef)
        env = dict(os.environ)
        if "group" in z.files:
            open(os.path.join(td, "meta"), "w").write("".join("%d\n" % x for x in z["group"]))
            env["FMX_META"] = os.path.join(td, "meta")
        for seed in range(201, 213):
            pre = os.path.join(td, "o%d" % seed)
            subprocess.run([HARNESS, "mcmc_gpu", trf, tef, str(z["task"]), "1", "1", str(int(z["k"])), str(int(z["iters"])),
                            repr(float(z["init_stdev"])), str(seed), pre], check=True, capture_output=True, env=env)
            p = np.fromfile(pre + ".pred_out.bin")
            preds.append(p)
            y = g.test_target.astype(np.float64)
            print(name, seed, "ours vs ref: corr %.4f rms %.4f | test rmse ours %.4f (fixture %.4f)" % (
                np.corrcoef(p, ref)[0, 1], np.sqrt(np.mean((p - ref) ** 2)), np.sqrt(np.mean((p - y) ** 2)), np.sqrt(np.mean((ref - y) ** 2))), flush=True)
    for i in range(1, len(preds)):
        print(name, "ours vs ours: corr %.4f rms %.4f" % (np.corrcoef(preds[0], preds[i])[0, 1], np.sqrt(np.mean((preds[0] - preds[i]) ** 2))))
