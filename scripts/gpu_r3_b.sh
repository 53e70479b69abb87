#!/bin/bash
# round 3: Criteo-shaped small-batch path: tests, bench, kernel stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_stability.py tests/test_gpu_parity.py tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -20 ) > $OUT/stability.log 2>&1
( timeout 400 python bench.py --workload criteo --rows 1048576 --steps 3 --warmup 1 2>/dev/null | grep "^{" ) > $OUT/bench_criteo.log 2>&1
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o criteo -- python $GRAFT_REPO_ROOT/bench.py --workload criteo --rows 262144 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 ) > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/criteo_kernel_stats.csv
rm -rf $OUT/prof
tail -12 $OUT/stability.log; cat $OUT/bench_criteo.log; head -12 $OUT/criteo_kernel_stats.csv
