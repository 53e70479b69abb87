#!/bin/bash
# round 4, third GPU call: what an agent-scope publish of S_e costs (variants sp1 / sp2), the generator test, counters of configs[1], Criteo rate
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python scripts/gpu_ab_variants.py sp1,sp2 2 fused ) > $OUT/ab_s_publish.txt 2>&1; tail -4 $OUT/ab_s_publish.txt
( timeout 600 python -m pytest tests/test_gpu_mcmc.py tests/test_gpu_zz_placement.py -q -m gpu -x 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python bench.py --workload criteo --rows 1048576 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" | cut -c1-200
cd /tmp
C2="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --features 10000000 --factors 32 --nnz 16 --rows 8388608 --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_c2_$tag -o c2 -- $C2 > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_trace -o c2 -- $C2 > $OUT/c2_under_rocprof.json 2>/dev/null
cp $OUT/c2_trace/*/c2_kernel_stats.csv $OUT/c2_kernel_stats.csv 2>/dev/null || cp $OUT/c2_trace/c2_kernel_stats.csv $OUT/c2_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT 40 2>/dev/null | grep "^==\|k_fused\|k_apply_seg" | cut -c1-150
head -5 $OUT/c2_kernel_stats.csv | cut -c1-50,150-250
find $OUT -name "*.csv" -size +3M -delete; rm -rf $OUT/c2_trace
