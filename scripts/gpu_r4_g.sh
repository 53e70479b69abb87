#!/bin/bash
# the k = 64 chain band + a soak of the whole fuzz file over other seeds (32-byte deferred records, early occurrence requests, unit streams)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 900 python -X faulthandler -m pytest tests/test_gpu_mcmc.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) > $OUT/mcmc.log 2>&1
tail -4 $OUT/mcmc.log
( FMX_FUZZ_SEEDS=200:260 timeout 1200 python -X faulthandler -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) > $OUT/fuzz_soak.log 2>&1
tail -4 $OUT/fuzz_soak.log
