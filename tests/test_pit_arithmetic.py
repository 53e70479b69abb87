"""the arithmetic of the parallel-in-time bias recurrence (k_scan_pit) on the CPU: Newton on the whole path + an affine prefix recurrence ends
where the serial chain (fm_sgd.h:34-37 summed per micro-chunk) ends, in a handful of iterations, for every micro-chunk incl. the reference's
own (1), both tasks, drifting and settled biases, ragged last chunks, a regularised bias"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from cpu_pit_check import newton, serial  # noqa: E402


@pytest.mark.parametrize("task", [0, 1])
@pytest.mark.parametrize("chunk", [1, 4, 32, 256, 2048])
@pytest.mark.parametrize("skew,w0,reg0", [(0.5, 0.0, 0.0), (0.85, 0.0, 0.0), (0.5, 1.5, 0.01)])
def test_newton_on_the_path_is_the_serial_chain(task, chunk, skew, w0, reg0):
    rng = np.random.default_rng(7 + chunk)
    n = 20000 + 37
    rest = (rng.standard_normal(n) * 0.3).astype(np.float32)
    if task == 1:
        y = np.where(rng.random(n) < skew, 1.0, -1.0).astype(np.float32)
    else:
        y = (rng.standard_normal(n) * 0.5 + (skew - 0.5)).astype(np.float32)      # the clamp [-1, 1] bites for some rows
    lr = min(0.01, 0.9 / (chunk * (1.0 if task == 0 else 0.25)))
    ws = serial(rest, y, chunk, lr, reg0, w0, task)
    wn, changes = newton(rest, y, chunk, lr, reg0, w0, task)
    if not changes[-1] < 5e-4:
        # nearly every prediction starts outside [min_target, max_target]: the clamped multiplier has derivative 0 there, the linearised chain
        # does not see the clamp release, and Newton does not settle -- the ONE case the device answers with its serial fallback (PIT_MAX_IT)
        assert task == 0 and w0 == 1.5 and len(changes) == 12, changes
        return
    assert len(changes) <= 8 or (task == 0 and w0 == 1.5), changes
    assert abs(ws - wn) <= 2e-6 * max(1.0, abs(ws)), (ws, wn, changes)
