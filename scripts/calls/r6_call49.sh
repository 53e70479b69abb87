#!/bin/bash
# round 6, GPU call 49: the new in-launch exchanges under a counter-collecting profiler (serialised dispatch): do they finish in normal time?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c49
mkdir -p $O
for s in small_one_rate seq_rate; do
T0=$(date +%s.%N)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$s -o p -- python scripts/$s.py > $O/$s.log 2>&1; rc=$?
echo "$s under --pmc FETCH_SIZE: rc=$rc $(echo "$(date +%s.%N) - $T0" | bc) s"
grep -E "examples/s" $O/$s.log
rm -rf $O/pmc_$s
done
