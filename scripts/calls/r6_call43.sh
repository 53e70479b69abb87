#!/bin/bash
# round 6, GPU call 43: the slots at SYSTEM scope (loads that never hit an L2): when do the owners see their tags?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c43
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_small_one.py -q -m gpu -x 2>&1 | tail -2
for dl in 0 32 64 96 128; do
echo "FMX_SMALL_DELAY=$dl"
FMX_SMALL_DELAY=$dl FMX_SMALL_TRACE=$O/trace_$dl.txt timeout 300 python scripts/small_one_rate.py 2>&1 | grep "one launch"
tail -2 $O/trace_$dl.txt | cut -c1-400
done
