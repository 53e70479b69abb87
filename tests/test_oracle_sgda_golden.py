"""CPU: pins the SGDA restatement (oracle/fm_oracle.c fmo_sgda_epoch) against the REAL reference's
fm_learn_sgd_element_adapt_reg (`-method sgda`).  Bar: bit-exact fp64 (parameters and learned regularisation)."""
import numpy as np
import pytest

from common import Golden
from conftest import golden_cases

CASES = [c for c in golden_cases() if c.startswith("sgda_")]


def val_data(g, O):
    z = g.z
    t = z["val_target"].copy()
    if g.task == 1:
        t = np.where(t <= 0, -1.0, 1.0).astype(np.float32)
    return O.Data(z["val_entries"], z["val_row_ptr"], t)


@pytest.mark.parametrize("name", CASES)
def test_sgda_bit_exact(oracle, name):
    O = oracle
    g = Golden(name)
    m = g.model(O, "init")
    m.reg0 = m.regw = m.regv = 0.0
    group = g.z["group"] if "group" in g.z.files else None            # `-meta` attribute groups
    st = O.sgda_learn(m, g.data(O, "train"), val_data(g, O), g.task, g.lr, g.min_target, g.max_target, g.iters, group)
    assert m.w0 == float(g.z["final_w0"])
    assert np.array_equal(m.w, g.z["final_w"])
    assert np.array_equal(m.v, g.z["final_v"])
    regs = g.z["regs"].reshape(st.num_groups, 1 + g.k)                # [G][1+k]: reg_w(g), reg_v(g,f)
    assert np.array_equal(st.reg_w, regs[:, 0])
    assert np.array_equal(st.reg_v[:, :g.k], regs[:, 1:])
    if group is not None:
        assert st.num_groups > 1 and np.unique(regs[:, 1:], axis=0).shape[0] > 1   # the groups really learn different values
    assert np.array_equal(O.predict_out(m, g.data(O, "test"), g.task, g.min_target, g.max_target), g.z["pred_out"])


@pytest.mark.parametrize("name", CASES)
def test_sgda_batch_rule_collapses_to_the_reference_at_batch_1(oracle, name):
    """the batch restatement (fmo_sgda_epoch_minibatch, what fmx_sgda_epoch_minibatch runs) with batch = chunk = 1 is the
    reference's loop: theta step, then one lambda step; not bit-exact only because the rule adds w0 to the row's sum last."""
    O = oracle
    g = Golden(name)
    if g.has_duplicate_ids():
        pytest.skip("rows with a repeated id: the batch rule uses batch-start parameters for both occurrences")
    m = g.model(O, "init")
    m.reg0 = m.regw = m.regv = 0.0
    group = g.z["group"] if "group" in g.z.files else None
    st = O.sgda_learn(m, g.data(O, "train"), val_data(g, O), g.task, g.lr, g.min_target, g.max_target, g.iters, group, batch=1, w0_chunk=1)
    np.testing.assert_allclose(m.w0, float(g.z["final_w0"]), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(m.w, g.z["final_w"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(m.v, g.z["final_v"], rtol=1e-10, atol=1e-13)
    regs = g.z["regs"].reshape(st.num_groups, 1 + g.k)
    np.testing.assert_allclose(st.reg_w, regs[:, 0], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(st.reg_v[:, :g.k], regs[:, 1:], rtol=1e-9, atol=1e-13)


def test_sgda_batch_rule_stays_near_the_online_result(oracle):
    """sanity of the restatement at a real batch size: same learned model quality, regularisation of the same size"""
    O = oracle
    g = Golden("sgda_reg_ml")
    res = []
    for batch in (None, 16):
        m = g.model(O, "init")
        m.reg0 = m.regw = m.regv = 0.0
        st = O.sgda_learn(m, g.data(O, "train"), val_data(g, O), g.task, g.lr, g.min_target, g.max_target, g.iters, None, batch=batch, w0_chunk=4)
        res.append((O.evaluate(m, g.data(O, "test"), g.task, g.min_target, g.max_target)[0], st.reg_w.copy(), st.reg_v.copy()))
    assert abs(res[0][0] - res[1][0]) < 0.03 * res[0][0]
    assert np.abs(res[1][2]).max() < 5 * max(np.abs(res[0][2]).max(), 1e-6) + 1e-3
