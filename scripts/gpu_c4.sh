#!/bin/bash
# BASELINE configs[4] shape (MCMC, k=128, 1e8 features) on ONE GPU (the table fits: 51 GB), and configs[3] (ALS k=64)
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-extras --method mcmc --features 100000000 --factors 128 --nnz 16 --rows 4194304 --steps 3 --warmup 1 2>&1 | grep "^{\|Error\|error" | cut -c1-700
timeout 300 python bench.py --no-cpu-baseline --no-extras --method mcmc --steps 4 --warmup 1 2>&1 | grep "^{\|Error\|error" | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
