#!/bin/bash
# round-end evidence (round 3): the whole GPU suite, rocprofv3 kernel stats of the default bench command and of the Criteo-shaped
# one, PMC byte counters in separate passes, the default bench line (CPU legs, extras), the Criteo-shaped line, two shards on one
# device as the driver types it.  Summaries are copied into profiles/r03_* by hand afterwards.
R=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${R}_profile
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- $B > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o bench -- $B --steps 3 --warmup 1 > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/criteo_trace -o criteo -- $B --workload criteo --rows 1048576 --steps 3 --warmup 1 > $OUT/criteo_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/als_trace -o als -- $B --method als --steps 3 --warmup 1 > $OUT/als_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT 8 > $OUT/pmc_summary.txt 2>&1
head -5 $OUT/bench_trace/bench_kernel_stats.csv | cut -c1-160
grep -A3 "^== pmc" $OUT/pmc_summary.txt | cut -c1-150
timeout 900 python bench.py 2>/dev/null | grep "^{" > $OUT/bench_default.json
cut -c1-300 $OUT/bench_default.json
timeout 400 python bench.py --workload criteo --rows 1048576 --steps 3 --warmup 1 2>/dev/null | grep "^{" > $OUT/bench_criteo.json
timeout 300 python bench.py --gpus 2 --same-device --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/bench_two_shards_one_device.json
cut -c1-200 $OUT/bench_criteo.json; grep -o '"phases_ms_per_batch.*' $OUT/bench_two_shards_one_device.json | cut -c1-200; cut -c1-200 $OUT/bench_two_shards_one_device.json
cp $OUT/bench_trace/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv; cp $OUT/criteo_trace/criteo_kernel_stats.csv $OUT/criteo_kernel_stats.csv; cp $OUT/als_trace/als_kernel_stats.csv $OUT/als_kernel_stats.csv
rm -rf $OUT/bench_trace $OUT/criteo_trace $OUT/als_trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
