#!/bin/bash
# round 6, GPU call 26: the abort inside fmx_create seen once in tests/test_gpu_placement.py::test_candidate_bound_is_honoured -- repeat with stderr visible
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c26
mkdir -p $O
for i in $(seq 1 14); do
timeout 600 python -m pytest tests/test_gpu_placement.py -q -m gpu -s -p no:faulthandler > $O/run_$i.txt 2>&1; rc=$?
echo "run $i rc=$rc $(grep -E 'passed|failed' $O/run_$i.txt | tail -1)"
if [ $rc -ne 0 ]; then tail -30 $O/run_$i.txt | cut -c1-300; fi
done
