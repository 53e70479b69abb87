#!/bin/bash
# round 6, GPU call 36: k_small_one A/B: write-through S stores (default) vs release fence (1), owners that do not wait (4: timing only), no owners (8: timing only)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_small_one.py -q -m gpu -x 2>&1 | tail -3
for f in 0 1 4 8 12; do echo "FMX_SMALL_FLAGS=$f"; FMX_SMALL_FLAGS=$f timeout 300 python scripts/small_one_rate.py 2>&1 | grep "one launch"; done
timeout 300 python scripts/small_one_rate.py 2>&1 | grep "two launches"
