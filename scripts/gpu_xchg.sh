#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/gpu_quick.sh "tests/test_gpu_group.py tests/test_gpu_sharded_driver.py tests/test_gpu_adapter.py"
for c in 1 4; do echo "== FMX_XCHG_CHUNKS=$c"; FMX_XCHG_CHUNKS=$c timeout 200 python bench.py --no-cpu-baseline --no-extras --force-sharded --mode minibatch --steps 6 2>/dev/null | grep "^{" | cut -c90-200; done
