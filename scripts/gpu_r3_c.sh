#!/bin/bash
# round 3: new test files (bench as typed, ingest, bench-batch oracle check), default bench with the als / mcmc extras
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_ingest.py tests/test_gpu_stability.py tests/test_gpu_fullsize.py -q -m gpu -s 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
( timeout 600 python bench.py 2>/dev/null | grep "^{" ) > $OUT/bench_default.json 2>&1
tail -30 $OUT/pytest.log; cat $OUT/bench_default.json
