#!/bin/bash
# round 6, GPU call 25: flakiness hunt on the final code -- fuzz seeds the suite does not use, then the whole suite three times
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c25
mkdir -p $O
FMX_FUZZ_SEEDS=4000:4060 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu > $O/fuzz.txt 2>&1; echo "fuzz rc=$? $(grep -E 'passed|failed' $O/fuzz.txt | tail -1)"; grep -E "^FAILED" $O/fuzz.txt | head
for i in 1 2 3; do
timeout 1500 python -m pytest tests -q -m gpu > $O/suite_$i.txt 2>&1; echo "suite $i rc=$? $(grep -E 'passed|failed' $O/suite_$i.txt | tail -1)"; grep -E "^FAILED" $O/suite_$i.txt | head -5
done
