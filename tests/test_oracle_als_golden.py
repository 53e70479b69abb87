"""CPU: pins the ALS restatement (oracle/fm_oracle_als.c) against the REAL reference's fm_learn_mcmc run with
do_sample = 0, do_multilevel = 0 (what `-method als` selects, libfm.cpp:135-139).  Bar: bit-exact fp64."""
import numpy as np
import pytest

from common import Golden
from conftest import golden_cases

CASES = [c for c in golden_cases() if c.startswith("als_")]


@pytest.mark.parametrize("name", CASES)
def test_als_bit_exact(oracle, name):
    O = oracle
    g = Golden(name)
    m = g.model(O, "init")
    tr, te = g.data(O, "train"), g.data(O, "test")
    if "group" in g.z.files:                                           # -meta + per-group lambdas (libfm.cpp:353-363)
        vl = np.repeat(g.z["v_lambda_g"][:, None], max(g.k, 1), axis=1)[:, :g.k]
        pred, metric = O.als_learn_groups(m, tr, te, g.task, g.iters, g.z["group"], g.z["w_lambda_g"], vl, g.min_target, g.max_target)
    else:
        pred, metric = O.als_learn(m, tr, te, g.task, g.iters, g.reg[1], g.reg[2], g.min_target, g.max_target)
    assert m.w0 == float(g.z["final_w0"])
    assert np.array_equal(m.w, g.z["final_w"])
    assert np.array_equal(m.v, g.z["final_v"])
    # fm_learn_mcmc::predict (fm_learn_mcmc.h:380-404): last iterate, clamped
    if g.task == 0:
        out = np.maximum(g.min_target, np.minimum(g.max_target, pred))
    else:
        out = np.maximum(0.0, np.minimum(1.0, pred))
    assert np.array_equal(out, g.z["pred_out"])
    assert np.isfinite(metric).all()
