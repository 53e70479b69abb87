#!/bin/bash
# round 6, GPU call 35: small batches as one launch per batch (k_small_one): parity cases, rate against the two launches
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c35
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_small_one.py -q -m gpu -x 2>&1 | tail -15
timeout 300 python scripts/small_one_rate.py 2>&1 | grep -v amdgpu.ids
