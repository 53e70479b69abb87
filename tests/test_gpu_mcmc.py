"""GPU: the Gibbs sampler (fm_learn_mcmc, do_sample = 1, do_multilevel = 1) -- STATISTICAL parity.

The reference draws from one sequential libc rand() stream with rejection loops (random.h); no parallel sampler
can reproduce it bit-for-bit, so the bar is: same model, same hyper-priors, same data -> the posterior-mean test
predictions of our chain agree with those of the reference's chain (fixtures produced by oracle/_ref/ref_harness
mcmc, 40 iterations): test RMSE / accuracy within 3 % and a prediction correlation inside the band the REFERENCE
shows against itself when only its -seed changes (measured with three other seeds: regression 0.986-0.990, rms
difference 0.09-0.11; classification 0.967-0.972, rms difference 0.075-0.082)."""
import io

import numpy as np
import pytest

from common import Golden

pytestmark = pytest.mark.gpu


def run_chain(g, oracle, seed):
    from libfm_amd import learner as L
    z = g.z
    fm = L.FMModel()
    fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = g.n, g.k, bool(g.k0), bool(g.k1)
    fm.w0, fm.w, fm.v = float(z["init_w0"]), z["init_w"].copy(), z["init_v"].copy()
    l = L.FMLearnMCMC()
    l.fm, l.task, l.num_iter, l.seed = fm, g.task, g.iters, seed
    l.min_target, l.max_target = g.min_target, g.max_target
    l.w_lambda = l.v_lambda = 0.0                              # libfm.cpp:329-335: no -regular -> lambdas start at 0
    l.out = io.StringIO()
    train = L.Data(z["train_entries"], z["train_row_ptr"], g.train_target)
    test = L.Data(z["test_entries"], z["test_row_ptr"], g.test_target)
    l.init()
    l.learn(train, test)
    p = l.predict(test)
    l.close()
    return p, l


def test_mcmc_regression_posterior_mean(oracle):
    g = Golden("mcmc_reg_ml")
    ref = g.z["pred_out"]
    y = g.test_target.astype(np.float64)
    p, l = run_chain(g, oracle, seed=3)
    rmse_ref = np.sqrt(np.mean((ref - y) ** 2))
    rmse = np.sqrt(np.mean((p - y) ** 2))
    assert abs(rmse - rmse_ref) < 0.03 * rmse_ref, (rmse, rmse_ref)
    assert np.corrcoef(p, ref)[0, 1] > 0.975                 # reference vs reference (other seeds): 0.986-0.990
    assert np.sqrt(np.mean((p - ref) ** 2)) < 0.14           # reference vs reference: 0.093-0.112
    assert all(np.isfinite(x["alpha"]) and x["alpha"] > 0 for x in l.log)


def test_mcmc_classification_posterior_mean(oracle):
    g = Golden("mcmc_cls_fields")
    ref = g.z["pred_out"]
    y = g.test_target
    p, l = run_chain(g, oracle, seed=5)
    acc_ref = np.mean((ref >= 0.5) == (y > 0))
    acc = np.mean((p >= 0.5) == (y > 0))
    assert abs(acc - acc_ref) < 0.03, (acc, acc_ref)
    assert np.corrcoef(p, ref)[0, 1] > 0.955                 # reference vs reference (other seeds): 0.967-0.972
    assert np.sqrt(np.mean((p - ref) ** 2)) < 0.10           # reference vs reference: 0.075-0.082
    assert (p >= 0).all() and (p <= 1).all()


def test_mcmc_two_seeds_differ_but_agree(oracle):
    g = Golden("mcmc_reg_ml")
    p1, _ = run_chain(g, oracle, seed=1)
    p2, _ = run_chain(g, oracle, seed=2)
    assert not np.array_equal(p1, p2)
    assert np.corrcoef(p1, p2)[0, 1] > 0.97
