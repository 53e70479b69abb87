#!/bin/bash
# round 6, GPU call 34: one-launch runs with the row_ptr trip gone; the time-out test; the sequential tests of the other files
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c34
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_safety.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_adapter.py -q -m gpu -k "conflict_free or long_runs or never_sees or sequential or trajectory or patched" -x 2>&1 | tail -8
timeout 300 python scripts/seq_rate.py 2>&1 | grep examples
