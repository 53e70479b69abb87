#!/bin/bash
# round 6, GPU call 22: is test_gpu_fuzz seed 16 flaky, and does it depend on the sequential kernel that ran before it?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "every_form" 2>&1 | grep -E "passed|failed" ; done
echo "--- FMX_SEQ_WG=0"
for i in 1 2 3; do FMX_SEQ_WG=0 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "every_form" 2>&1 | grep -E "passed|failed" ; done
echo "--- FMX_SEQ_ROWS=0"
for i in 1 2 3; do FMX_SEQ_ROWS=0 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "every_form" 2>&1 | grep -E "passed|failed" ; done
