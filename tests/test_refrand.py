"""CPU: `-seed` parity -- the host side reproduces the reference's initial parameters bit for bit (same libc rand()
stream, same Leva sampler, same fill order) for the seeds of the golden fixtures."""
import numpy as np
import pytest

from common import Golden
from conftest import golden_cases


@pytest.mark.parametrize("name", [c for c in golden_cases() if c.startswith(("sgd_", "als_"))])
def test_seeded_init_is_bit_identical(name):
    from libfm_amd import refrand as R
    g = Golden(name)
    z = g.z
    R.srand(int(z["seed"]))
    v = R.init_v(g.k, g.n, 0.0, float(z["init_stdev"]))
    assert np.array_equal(v, z["init_v"])
    if name.startswith("als_"):                       # mcmc/als also randomise w afterwards (libfm.cpp:283)
        w = R.init_w_normal(g.n, 0.0, float(z["init_stdev"]))
        assert np.array_equal(w, z["init_w"])
    else:
        assert not z["init_w"].any()
