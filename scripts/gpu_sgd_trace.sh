#!/bin/bash
# kernel trace of the default bench command (quick look at the three kernels of a batch)
OUT=$GRAFT_REPO_ROOT/gpurun_out/sgd_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 8 "$@" 2>/dev/null | grep "^{" | cut -c90-200
python - <<'P'
import csv,os
for r in list(csv.reader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sgd_trace/t/bench_kernel_stats.csv')))[1:5]:
    print(r[0][:48], r[1], r[3], r[4])
P
