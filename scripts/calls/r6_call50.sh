#!/bin/bash
# round 6, GPU call 50: the one-launch forms at k <= 32 (several rows per wave-wide load): parity, rates
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_small_one.py tests/test_gpu_parity.py tests/test_gpu_safety.py -q -m gpu -x -k "one_launch or conflict_free or long_runs or never_sees or gives_up or weight_side" 2>&1 | tail -6
timeout 300 python scripts/seq_rate.py 2>&1 | grep examples
FMX_SEQ_RUNS_ONE=0 timeout 300 python scripts/seq_rate.py 2>&1 | grep "k=8"
