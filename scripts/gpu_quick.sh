#!/bin/bash
# quick GPU check of a list of test files / -k expression:  gpurun -- 'bash scripts/gpu_quick.sh "<pytest args>"'
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest $1 -q -m gpu --maxfail=10 2>&1 | tail -60 ) > $OUT/pytest.log 2>&1
tail -60 $OUT/pytest.log
