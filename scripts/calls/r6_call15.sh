#!/bin/bash
# round 6, GPU call 15: rows of k rounded up to 16 floats (not to the power of two): the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c15
mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -15 $O/pytest_gpu.txt
