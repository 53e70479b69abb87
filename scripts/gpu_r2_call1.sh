#!/bin/bash
# round 2, GPU call 1: correctness of the one-pass minibatch form + first measurements + the w-gather micro-benchmark
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or hogwild or odd_k or kilo" 2>&1 | tail -15 ) > $OUT/pytest_new.log 2>&1
tail -3 $OUT/pytest_new.log
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "fused or hogwild" 2>&1 | tail -15 ) > $OUT/pytest_full.log 2>&1
tail -3 $OUT/pytest_full.log
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
run() { echo "== $*" >> $OUT/bench_variants.log; ( "$@" 2>&1 | grep "^{" ) >> $OUT/bench_variants.log; }
run $B
run $B --bias-lag 3
run $B --bias-lag 1
run $B --batch 131072
run $B --batch 524288
run $B --mode hogwild
run env FMX_W_ALLOC=1 $B
run env FMX_W_ALLOC=1 $B --mode hogwild
run env FMX_W_ALLOC=2 $B --mode hogwild
run env FMX_SCAN_CU=1 $B
run env FMX_SCAN_CU=1 $B --mode hogwild
grep -o '^== .*\|"value": [0-9.]*\|"frac": [0-9.]*\|deferred_features_per_example": [0-9.]*' $OUT/bench_variants.log | paste -sd' ' | sed 's/== /\n== /g'
# micro-benchmark: timing table, then byte counters in separate passes
( timeout 300 scripts/ubench/w_gather ) > $OUT/w_gather_timing.txt 2>&1
cat $OUT/w_gather_timing.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for kind in 0 1 2; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/wg_${c}_kind$kind -o wg -- $GRAFT_REPO_ROOT/scripts/ubench/w_gather 800000000 67108864 $kind > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT 16 > $OUT/w_gather_pmc.txt 2>&1
grep -v "no counter csv" $OUT/w_gather_pmc.txt | cut -c1-150
