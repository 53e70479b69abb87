#!/bin/bash
# round 6, GPU call 39: k_small_one keeping the weight side stream; the give-up test; rate
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_small_one.py tests/test_gpu_safety.py tests/test_gpu_parity.py -q -m gpu -x -k "one_launch or weight_side or gives_up or never_sees" 2>&1 | tail -5
timeout 300 python scripts/small_one_rate.py 2>&1 | grep launch
