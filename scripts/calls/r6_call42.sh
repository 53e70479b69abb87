#!/bin/bash
# round 6, GPU call 42: owners that start polling later (FMX_SMALL_DELAY x 64 cycles): do the polls slow the examples' gathers down?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c42
mkdir -p $O
for dl in 0 64 128 192 256; do
echo "FMX_SMALL_DELAY=$dl"
FMX_SMALL_DELAY=$dl FMX_SMALL_TRACE=$O/trace_$dl.txt timeout 300 python scripts/small_one_rate.py 2>&1 | grep "one launch"
tail -2 $O/trace_$dl.txt | cut -c1-400
done
