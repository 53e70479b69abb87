#!/bin/bash
# round 6, GPU call 30: where the time of a conflict-free run goes (kernel stats of scripts/seq_rate.py)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c30
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o seq -- python scripts/seq_rate.py > $O/prof.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/seq_kernel_stats.csv \;
rm -rf $O/kt
head -25 $O/seq_kernel_stats.csv | cut -c1-200
tail -4 $O/prof.log
