#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_stability.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
for i in 1 2; do ( timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" ) >> $OUT/bench.json 2>&1; done
tail -25 $OUT/pytest.log; cut -c1-120 $OUT/bench.json; grep -o '"roofline.*' $OUT/bench.json | cut -c1-400
