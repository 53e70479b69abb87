#!/bin/bash
# BASELINE configs[1] shape (n = 1e7, k = 32, 16 nnz/row) on one GPU: rate + kernel stats (+ optional bench args)
OUT=$GRAFT_REPO_ROOT/gpurun_out/c2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o c2 -- python $GRAFT_REPO_ROOT/bench.py --features 10000000 --factors 32 --nnz 16 --rows 8388608 --no-cpu-baseline --no-extras --steps 6 "$@" 2>/dev/null | grep "^{" | cut -c1-200
python - <<'P'
import csv,os
for r in list(csv.reader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/c2/t/c2_kernel_stats.csv')))[1:5]:
    print(r[0][:60], r[1], r[3][:9], r[4])
P
cp $OUT/t/c2_kernel_stats.csv $OUT/c2_kernel_stats.csv; rm -rf $OUT/t
