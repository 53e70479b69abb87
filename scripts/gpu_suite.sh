#!/bin/bash
# the whole GPU suite + the default bench line:  gpurun -- 'bash scripts/gpu_suite.sh <tag>'
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-suite}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -60 ) > $OUT/pytest.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" ) > $OUT/bench.json 2>&1
tail -25 $OUT/pytest.log; cut -c1-400 $OUT/bench.json
