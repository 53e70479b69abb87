#!/bin/bash
# ALS / MCMC after the level-ordered column records: tests, configs[3] and configs[4] shapes with kernel stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03als
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_mcmc.py tests/test_gpu_relations.py tests/test_gpu_group.py tests/test_gpu_fuzz.py tests/test_gpu_adapter.py -q -m gpu 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 300 python bench.py --method als --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep "^{" | cut -c1-230
bash scripts/gpu_c5_trace.sh
