"""Helpers shared by the CPU and GPU parity tests."""
import os

import numpy as np

from conftest import GOLDEN_DIR


class Golden:
    """One tests/golden/*.npz fixture (inputs, config and the REAL reference's outputs)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.z = z
        self.task = 0 if str(z["task"]) == "r" else 1
        self.k0, self.k1, self.k = int(z["k0"]), int(z["k1"]), int(z["k"])
        self.iters, self.lr = int(z["iters"]), float(z["lr"])
        self.reg = [float(x) for x in z["reg"]]
        self.n = int(z["n"])
        self.train_target = z["train_target"].copy()
        self.test_target = z["test_target"].copy()
        # min/max target come from the TRAIN set as parsed (libfm.cpp:295-296, Data.h:201-202) ...
        self.min_target = float(self.train_target.min())
        self.max_target = float(self.train_target.max())
        # ... and for classification main() rewrites targets to +-1 before learn (libfm.cpp:302-306)
        if self.task == 1:
            self.train_target = np.where(self.train_target <= 0, -1.0, 1.0).astype(np.float32)
            self.test_target = np.where(self.test_target <= 0, -1.0, 1.0).astype(np.float32)

    def data(self, O, which):
        z = self.z
        tgt = self.train_target if which == "train" else self.test_target
        return O.Data(z[which + "_entries"], z[which + "_row_ptr"], tgt)

    def model(self, O, which="init"):
        z = self.z
        m = O.Model(self.n, self.k, self.k0, self.k1, *self.reg)
        m.w0 = float(z[which + "_w0"])
        m.w[:] = z[which + "_w"]
        m.v[:] = z[which + "_v"]
        return m

    def has_duplicate_ids(self):
        z = self.z
        rp = z["train_row_ptr"].astype(np.int64)
        ids = z["train_entries"]["id"]
        for r in range(len(rp) - 1):
            row = ids[rp[r]:rp[r + 1]]
            if len(np.unique(row)) != len(row):
                return True
        return False
