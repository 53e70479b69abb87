#!/bin/bash
# round 4, first GPU call: the whole GPU suite (new: tests/test_gpu_configs.py, the hand-off tests), hand-off vs events A/B over batch
# sizes, the default bench line with the new extra keys, kernel stats of the headline, PMC of the ALS sweep's kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
( timeout 300 python scripts/gpu_ab_handoff.py 262144,131072,65536 ) > $OUT/ab_handoff.txt 2>&1
cat $OUT/ab_handoff.txt
( timeout 300 python scripts/gpu_ab_handoff.py 262144,131072,65536 10000000 32 16 ) > $OUT/ab_handoff_c2.txt 2>&1
cat $OUT/ab_handoff_c2.txt
timeout 900 python bench.py 2>$OUT/bench_default.err | grep "^{" > $OUT/bench_default.json
cut -c1-600 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- $B > $OUT/bench_under_rocprof.json 2>/dev/null
cp $OUT/bench_trace/*/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null || cp $OUT/bench_trace/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv
head -6 $OUT/bench_kernel_stats.csv | cut -c1-170
A="$B --method als --steps 2 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_als_$tag -o als -- $A > /dev/null 2>$OUT/pmc_als_$tag.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/als_trace -o als -- $A > $OUT/als_under_rocprof.json 2>/dev/null
cp $OUT/als_trace/*/als_kernel_stats.csv $OUT/als_kernel_stats.csv 2>/dev/null || cp $OUT/als_trace/als_kernel_stats.csv $OUT/als_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT 12 > $OUT/pmc_summary.txt 2>&1
grep -A8 "^== pmc" $OUT/pmc_summary.txt | cut -c1-160
find $OUT -name "*.csv" -size +3M -delete
rm -rf $OUT/bench_trace $OUT/als_trace
du -sh $OUT
