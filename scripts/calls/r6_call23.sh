#!/bin/bash
# round 6, GPU call 23: how often does the one-off fuzz failure of call 21 come back in the same command?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c23
mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adapter.py tests/test_gpu_fuzz.py -q -m gpu > $O/run_$i.txt 2>&1; echo "run $i rc=$? $(grep -E 'passed|failed' $O/run_$i.txt | tail -1) $(grep -E '^FAILED' $O/run_$i.txt | head -3)"
done
