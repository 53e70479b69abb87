#!/bin/bash
# round-end evidence: rocprofv3 kernel stats of the default bench command + PMC byte counters (separate passes)
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -o cal -- python $GRAFT_REPO_ROOT/scripts/gpu_calib.py > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/als_trace -o als -- python $GRAFT_REPO_ROOT/scripts/gpu_als_probe.py > $OUT/als_probe.log 2>&1
cd $GRAFT_REPO_ROOT
head -12 $OUT/als_trace/als_kernel_stats.csv | cut -c1-200
cat $OUT/als_probe.log | tail -2
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
head -8 $OUT/bench_trace/bench_kernel_stats.csv | cut -c1-220
grep -A2 "^== pmc\|^== cal" $OUT/pmc_summary.txt | cut -c1-200
python bench.py 2>/dev/null | grep "^{" > $OUT/bench_default.json
cat $OUT/bench_default.json
