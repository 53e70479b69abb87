#!/bin/bash
# round 6, GPU call 21: the reference's own trajectory a ROW at a time (k_sequential_rows): parity tests that run FMX_SGD_SEQUENTIAL, and its rate
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c21
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adapter.py tests/test_gpu_fuzz.py -q -m gpu > $O/pytest.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/pytest.txt | tail -3; grep -E "^FAILED" $O/pytest.txt | head -10
python scripts/seq_rate.py 2>&1 | grep examples
FMX_SEQ_ROWS=0 python scripts/seq_rate.py 2>&1 | grep examples | head -1
