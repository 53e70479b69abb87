"""CPU: which batch the library runs the batch rule with (fmx_batch_rule, host arithmetic): an explicit batch is
honoured and judged, batch 0 is 262144 cut to the largest power of two with learn_rate * curvature * batch * collision_mass <= 1."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import build, capi
    build.build()
    return capi


def test_default_is_kept_when_the_rows_are_sparse(capi):
    bi = capi.batch_rule(capi.TASK_CLASSIFICATION, 0.01, 1.0e-5)          # the bench's uniform rows
    assert bi.batch == 262144 and bi.status == 0
    assert abs(bi.batch_gain - 0.01 * 0.25 * 262144 * 1.0e-5) < 1e-12
    bi = capi.batch_rule(capi.TASK_REGRESSION, 0.01, 0.0)
    assert bi.batch == 262144 and bi.status == 0 and bi.batch_gain == 0.0


def test_cut_to_a_power_of_two_below_gain_one(capi):
    bi = capi.batch_rule(capi.TASK_CLASSIFICATION, 0.01, 0.767)           # the Criteo-shaped rows of the bench
    assert bi.batch == 512 and (bi.status & capi.STAT_BATCH_CUT) and not (bi.status & capi.STAT_UNSTABLE)
    assert 0.5 < bi.batch_gain <= 1.0
    rng = np.random.default_rng(3)
    for _ in range(500):
        task = int(rng.integers(0, 2))
        lr = float(10 ** rng.uniform(-4, -0.5))
        c = float(10 ** rng.uniform(-7, 2))
        bi = capi.batch_rule(task, lr, c)
        curv = 1.0 if task == capi.TASK_REGRESSION else 0.25
        assert bi.batch & (bi.batch - 1) == 0 and 32 <= bi.batch <= 262144
        gain = lr * curv * bi.batch * c
        assert abs(bi.batch_gain - gain) <= 1e-9 * max(1.0, gain)
        if bi.batch > 32:
            assert gain <= 1.0 + 1e-12                                      # stable at the chosen batch ...
        if bi.batch < 262144 and bi.batch > 32:
            assert 2 * gain > 1.0 and (bi.status & capi.STAT_BATCH_CUT)     # ... and the next power of two would not be
        if bi.batch == 32 and gain > 2.0:
            assert bi.status & capi.STAT_UNSTABLE                           # rows that dense are SEQUENTIAL's case: the status says so


def test_explicit_batch_is_honoured_and_judged(capi):
    bi = capi.batch_rule(capi.TASK_REGRESSION, 0.01, 0.5, 100000)
    assert bi.batch == 100000 and bi.batch_gain == pytest.approx(500.0) and (bi.status & capi.STAT_UNSTABLE)
    assert not (bi.status & capi.STAT_BATCH_CUT)
    bi = capi.batch_rule(capi.TASK_REGRESSION, 0.01, 0.5, 100)
    assert bi.batch == 100 and bi.status == 0


def test_bad_arguments(capi):
    with pytest.raises(capi.FmxError):
        capi.batch_rule(7, 0.01, 0.1)
    with pytest.raises(capi.FmxError):
        capi.batch_rule(0, -1.0, 0.1)


def test_default_micro_chunk_of_the_bias_recurrence(capi):
    """fmx_default_w0_chunk (host arithmetic): the largest power of two <= FMX_W0_CHUNK_CAP = 32 with learn_rate * chunk * curvature <= 1
    (curvature 1 regression, 1/4 classification).  The cap -- 256 until ABI 5 -- sets how finely the rule follows the reference's per-example
    bias step (fm_sgd.h:34-37); the stability bound sets everything below it."""
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fmx.h")).read()
    cap = int(re.search(r"#define\s+FMX_W0_CHUNK_CAP\s+(\d+)", hdr).group(1))
    assert cap == 32
    assert capi.default_w0_chunk(0.01, capi.TASK_CLASSIFICATION) == 32           # the bench: 1 / (0.01 * 1/4) = 400 -> capped
    assert capi.default_w0_chunk(0.01, capi.TASK_REGRESSION) == 32               # 100 -> capped
    assert capi.default_w0_chunk(0.1, capi.TASK_REGRESSION) == 8                 # 10 -> 8
    assert capi.default_w0_chunk(0.2, capi.TASK_CLASSIFICATION) == 16            # 20 -> 16
    assert capi.default_w0_chunk(2.0, capi.TASK_REGRESSION) == 1                 # never below the reference's own step
    assert capi.default_w0_chunk(0.0, capi.TASK_REGRESSION) == cap
    rng = np.random.default_rng(11)
    for _ in range(300):
        task = int(rng.integers(0, 2))
        lr = float(10 ** rng.uniform(-4, 0.5))
        c = capi.default_w0_chunk(lr, task)
        curv = 1.0 if task == capi.TASK_REGRESSION else 0.25
        assert c >= 1 and c & (c - 1) == 0 and c <= cap
        assert c == 1 or lr * c * curv <= 1.0 + 1e-12
        assert c == cap or lr * (2 * c) * curv > 1.0
