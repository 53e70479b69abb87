"""CPU: block-structured data (`-relation`).  The reference's ALS never builds the joined rows: it sweeps per-block
caches (fm_learn_mcmc.h:478-527, draw_w_rel :734-790, draw_v_rel :849-909).  Those caches compute, in a different
summation order, exactly the coordinate updates of the FLAT design matrix (main entries followed by the mapped row of
every block, block attribute ids shifted by attr_offset).  This test pins that claim: the oracle's flat ALS on the
expanded rows must reproduce the REAL reference's block-structured run (fixtures from oracle/_ref/ref_harness with
FMX_RELATIONS) to accumulated fp64 rounding (observed <= 4e-8 relative after 3-4 sweeps; bar 1e-6 -- not bit-exact because
the order of the sums differs and the block caches are updated incrementally)."""
import numpy as np
import pytest

import datagen
from common import Golden
from conftest import golden_cases

CASES = [c for c in golden_cases() if c.startswith("rel_als_")]


def flat(g, O, which):
    z = g.z
    blocks = [(z["rel%d_entries" % i], z["rel%d_row_ptr" % i], int(z["rel%d_num_feature" % i])) for i in range(int(z["n_relations"]))]
    maps = [z["rel%d_%s" % (i, which)] for i in range(int(z["n_relations"]))]
    ent, rp, offs = datagen.expand_blocks(z[which + "_entries"], z[which + "_row_ptr"], blocks, maps, int(z["n_main"]))
    tgt = g.train_target if which == "train" else g.test_target
    return O.Data(ent, rp, tgt), offs


@pytest.mark.parametrize("name", CASES)
def test_flat_als_equals_block_structured_reference(oracle, name):
    O = oracle
    g = Golden(name)
    m = g.model(O, "init")
    tr, offs = flat(g, O, "train")
    te, _ = flat(g, O, "test")
    assert offs[0] == int(g.z["n_main"]) and g.n == offs[-1] + int(g.z["rel%d_num_feature" % (len(offs) - 1)])
    if "group" in g.z.files:
        vl = np.repeat(g.z["v_lambda_g"][:, None], g.k, axis=1)
        pred, _ = O.als_learn_groups(m, tr, te, g.task, g.iters, g.z["group"], g.z["w_lambda_g"], vl, g.min_target, g.max_target)
    else:
        pred, _ = O.als_learn(m, tr, te, g.task, g.iters, g.reg[1], g.reg[2], g.min_target, g.max_target)
    assert abs(m.w0 - float(g.z["final_w0"])) <= 1e-6 * max(1.0, abs(m.w0))
    np.testing.assert_allclose(m.w, g.z["final_w"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(m.v, g.z["final_v"], rtol=1e-6, atol=1e-9)
    out = np.clip(pred, g.min_target, g.max_target) if g.task == 0 else np.clip(pred, 0.0, 1.0)
    np.testing.assert_allclose(out, g.z["pred_out"], rtol=1e-6, atol=1e-9)
