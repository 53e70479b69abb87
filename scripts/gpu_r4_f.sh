#!/bin/bash
# the GPU suite again (after the ALS fix) + a soak of the large-batch fuzz over other seeds
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 1500 python -X faulthandler -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
( FMX_FUZZ_SEEDS=100:124 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "large_batches" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > $OUT/fuzz_soak.log 2>&1
tail -3 $OUT/fuzz_soak.log
