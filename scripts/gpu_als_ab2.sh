#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/als_ab
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o als -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --method als --steps 3 --warmup 1 2>/dev/null | grep "^{" | cut -c90-200
cut -c1-60,150-400 $OUT/trace2/als_kernel_stats.csv | head -5
