#!/bin/bash
# round 5, GPU call 10: k_als_shadow (whole rows per wavefront) and k_als_unseen_v (priors hoisted): parity, then the ALS / MCMC bench legs
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c10
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_als.py tests/test_gpu_mcmc.py tests/test_gpu_relations.py tests/test_gpu_ingest.py -q -m gpu > $O/pytest_als.log 2>&1
grep -E "passed|failed" $O/pytest_als.log | tail -2
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_adapter.py -q -m gpu -k "als or mcmc or chain or config3 or config4 or sweep or block" > $O/pytest_als2.log 2>&1
grep -E "passed|failed" $O/pytest_als2.log | tail -2
timeout 600 python bench.py --method mcmc --features 100000000 --factors 128 --nnz 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/mcmc_c5.json 2> $O/mcmc_c5.err
timeout 300 python bench.py --method als --steps 2 --warmup 1 --no-cpu-baseline > $O/als.json 2>> $O/mcmc_c5.err
timeout 300 python bench.py --method mcmc --steps 2 --warmup 1 --no-cpu-baseline > $O/mcmc.json 2>> $O/mcmc_c5.err
for f in mcmc_c5 als mcmc; do python -c "
import json; o=json.load(open('$O/$f.json')); print('$f', o['value'], o['ms_per_step'], o['roofline']['frac'])"; done
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python bench.py --method mcmc --features 100000000 --factors 128 --nnz 16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv; rm -rf $O/prof_c5
head -9 $O/c5_kernel_stats.csv | cut -c1-70,200-330
