#!/bin/bash
# round-end evidence (round 2): the whole GPU suite, rocprofv3 kernel stats of the default bench command, PMC byte counters in
# separate passes (SGD and ALS), ALS kernel stats in the split and in the fused form of the draws, and the default bench lines
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${R}_profile
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -q -m gpu --maxfail=15 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- $B > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o bench -- $B --steps 3 --warmup 1 > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/als_trace -o als -- $B --method als --steps 3 --warmup 1 > $OUT/als_under_rocprof.json 2>/dev/null
FMX_ALS_SPLIT_MIN=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/als_fused_trace -o als -- $B --method als --steps 3 --warmup 1 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_als_$c -o als -- $B --method als --steps 2 --warmup 1 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT 8 > $OUT/pmc_summary.txt 2>&1
head -5 $OUT/bench_trace/bench_kernel_stats.csv | cut -c1-160
grep -A3 "^== pmc" $OUT/pmc_summary.txt | cut -c1-150
timeout 600 python bench.py 2>/dev/null | grep "^{" > $OUT/bench_default.json
cat $OUT/bench_default.json
timeout 200 python bench.py --no-cpu-baseline --no-extras --method als 2>/dev/null | grep "^{" > $OUT/bench_als.json
timeout 200 python bench.py --no-cpu-baseline --no-extras --force-sharded --mode minibatch 2>/dev/null | grep "^{" > $OUT/bench_one_rank_rccl.json
cut -c1-300 $OUT/bench_als.json $OUT/bench_one_rank_rccl.json
