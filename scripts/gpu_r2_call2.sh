#!/bin/bash
# round 2, GPU call 2: the whole GPU suite, the merged collision pass, ALS shadow A/B, w-gather micro-benchmark (short timeouts:
# a faulting micro-benchmark must not eat the budget again)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2"
run() { echo "== $*" >> $OUT/bench_variants.log; ( timeout 200 "$@" 2>&1 | grep "^{" ) >> $OUT/bench_variants.log; }
run $B
run env FMX_FUSED_SEPARATE_PASS=1 $B
run $B --mode hogwild
run $B --bias-lag 3
run $B --batch 524288
run $B --batch 131072
run $B --method als --steps 3 --warmup 1
run env FMX_ALS_NO_SHADOW=1 $B --method als --steps 3 --warmup 1
grep -o '^== .*\|"value": [0-9.]*\|"frac": [0-9.]*\|deferred_features_per_example": [0-9.]*\|avg_sweep_ms": [0-9.]*' $OUT/bench_variants.log | paste -sd' ' | sed 's/== /\n== /g'
( timeout 120 scripts/ubench/w_gather ) > $OUT/w_gather_timing.txt 2>&1
cat $OUT/w_gather_timing.txt | tr '\r' '\n' | grep -v "\.\.\.$" | tail -40
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for kind in 0 1; do
    timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/wg_${c}_kind$kind -o wg -- $GRAFT_REPO_ROOT/scripts/ubench/w_gather 800000000 67108864 $kind > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT 16 > $OUT/w_gather_pmc.txt 2>&1
grep -v "no counter csv" $OUT/w_gather_pmc.txt | cut -c1-150
