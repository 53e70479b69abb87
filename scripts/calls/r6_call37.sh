#!/bin/bash
# round 6, GPU call 37: k_small_one with the recurrence one launch behind (lag >= 2) and the multiplier published behind the own updates; A/B: 1 = before them, 8 = no owners (timing only)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_small_one.py -q -m gpu -x 2>&1 | tail -3
for f in 0 1 8; do echo "FMX_SMALL_FLAGS=$f"; FMX_SMALL_FLAGS=$f timeout 300 python scripts/small_one_rate.py 2>&1 | grep "one launch"; done
timeout 300 python scripts/small_one_rate.py 2>&1 | grep "two launches"
