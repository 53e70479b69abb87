#!/bin/bash
# round 3, first GPU pass: new stability tests, the whole GPU suite, default bench, Criteo-shaped bench
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_stability.py -q -m gpu 2>&1 | tail -40 ) > $OUT/stability.log 2>&1
( timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_stability.py 2>&1 | tail -80 ) > $OUT/pytest.log 2>&1
( timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -3 ) > $OUT/bench_default.log 2>&1
( timeout 300 python bench.py --workload criteo --rows 1048576 --steps 3 --warmup 1 2>&1 | tail -3 ) > $OUT/bench_criteo.log 2>&1
( timeout 300 python bench.py --workload criteo --rows 1048576 --steps 3 --warmup 1 --bias-lag 1 2>&1 | tail -3 ) > $OUT/bench_criteo_lag1.log 2>&1
tail -40 $OUT/stability.log; tail -30 $OUT/pytest.log; cat $OUT/bench_default.log $OUT/bench_criteo.log $OUT/bench_criteo_lag1.log
