#!/bin/bash
# round 2, GPU call 3: the whole GPU suite (all failures listed), merged vs separate collision pass
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -q -m gpu --maxfail=12 --ignore=tests/test_gpu_fullsize.py 2>&1 | tail -60 ) > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --maxfail=5 2>&1 | tail -40 ) > $OUT/pytest_full.log 2>&1
tail -3 $OUT/pytest_full.log
B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2"
run() { echo "== $*" >> $OUT/bench_variants.log; ( timeout 200 "$@" 2>&1 | grep "^{" ) >> $OUT/bench_variants.log; }
run $B
run env FMX_FUSED_SEPARATE_PASS=1 $B
run $B --mode hogwild
run $B --batch 524288
run env FMX_FUSED_SEPARATE_PASS=1 $B --batch 524288
run $B --batch 131072
grep -o '^== .*\|"value": [0-9.]*\|"frac": [0-9.]*\|deferred_features_per_example": [0-9.]*\|avg_sweep_ms": [0-9.]*' $OUT/bench_variants.log | paste -sd' ' | sed 's/== /\n== /g'
