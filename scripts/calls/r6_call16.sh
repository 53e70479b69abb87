#!/bin/bash
# round 6, GPU call 16: factor counts up to 1024 (KP 512 / 1024): the tests that take k
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c16
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_als.py -q -m gpu -k "odd_k or register_paths or wide_rows" > $O/pytest_k.txt 2>&1; echo "rc=$?"; tail -30 $O/pytest_k.txt
