#!/bin/bash
# BASELINE configs[4] shape on one GPU (MCMC, k = 128, n = 1e8, 16 nnz/row): sweep time + kernel stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o c5 -- python $GRAFT_REPO_ROOT/bench.py --method mcmc --features 100000000 --factors 128 --nnz 16 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" | cut -c1-260
python - <<'P'
import csv,os
for r in list(csv.reader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/c5/t/c5_kernel_stats.csv')))[1:9]:
    print(r[0][:60], r[1], r[2], r[3][:9], r[4])
P
cp $OUT/t/c5_kernel_stats.csv $OUT/c5_kernel_stats.csv; rm -rf $OUT/t
