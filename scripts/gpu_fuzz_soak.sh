#!/bin/bash
# soak run of the seeded fuzz tests over seeds the suite does not use.  usage: gpurun -- 'bash scripts/gpu_fuzz_soak.sh 1000:1150'
mkdir -p gpurun_out
FMX_FUZZ_SEEDS=${1:-1000:1100} timeout ${2:-1200} python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 | tee gpurun_out/fuzz_soak.txt
