#!/bin/bash
# round 6, GPU call 38: k_small_one as the default for small batches: rate, then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c38
mkdir -p $O
timeout 300 python scripts/small_one_rate.py 2>&1 | grep launch
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -5 $O/pytest_gpu.txt
