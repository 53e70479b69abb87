#!/bin/bash
# A/B of the ALS sweep: fused draws vs split step (FMX_ALS_SPLIT_MIN), with a kernel trace of the split form
OUT=$GRAFT_REPO_ROOT/gpurun_out/als_ab
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_als.py -q -m gpu 2>&1 | tail -5 ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for m in 0 65536; do
  echo "== FMX_ALS_SPLIT_MIN=$m"
  FMX_ALS_SPLIT_MIN=$m timeout 300 python bench.py --no-cpu-baseline --no-extras --method als --steps 4 --warmup 1 2>/dev/null | grep "^{" | tee $OUT/bench_als_$m.json | cut -c1-260
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o als -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --method als --steps 3 --warmup 1 > /dev/null 2>&1
cut -c1-150 $OUT/trace/als_kernel_stats.csv | head -8
