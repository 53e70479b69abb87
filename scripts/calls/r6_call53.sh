#!/bin/bash
# round 6, GPU call 53: k_small_one on batches of 2048 / 4096 rows (FMX_SMALL_MAX; a smaller learning rate passes a larger batch): parity, rate
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
FMX_SMALL_MAX=4096 timeout 900 python -m pytest tests/test_gpu_small_one.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
for lr in 0.0025 0.00125; do
echo "LR=$lr"; LR=$lr FMX_SMALL_MAX=4096 timeout 300 python scripts/small_one_rate.py 2>&1 | grep launch
done
