#!/bin/bash
# round 2, GPU call 5: deferred-feature pass parallelism A/B, the library's own comm path with one rank, failed tests again
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_parity.py -q -m gpu --maxfail=10 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2"
run() { echo "== $*" >> $OUT/bench_variants.log; ( timeout 200 "$@" 2>&1 | grep "^{" ) >> $OUT/bench_variants.log; }
run $B
run env FMX_P2_SPW=8 $B
run env FMX_P2_SPW=32 $B
run env FMX_P2_SPW=64 $B
run $B --mode hogwild
run $B --batch 524288
run $B --batch 131072
run env FMX_FUSED_MERGE=1 $B
run $B --force-sharded --mode minibatch
run $B --force-sharded --mode minibatch --driver torch
run $B --force-sharded --mode minibatch --pipeline
grep -o '^== .*\|"value": [0-9.]*\|"frac": [0-9.]*\|deferred_features_per_example": [0-9.]*' $OUT/bench_variants.log | paste -sd' ' | sed 's/== /\n== /g'
grep -c "^{" $OUT/bench_variants.log
