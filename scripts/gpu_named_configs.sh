#!/bin/bash
# one line per named configuration of BASELINE.json (1 GPU): configs[1] C2, configs[2] C3 shape, configs[3] ALS, configs[4] MCMC shape
OUT=$GRAFT_REPO_ROOT/gpurun_out/named
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
echo "# configs[1]: synthetic 1e7 features, k=32, nnz=16, SGD (hogwild bench line)"
python bench.py --no-cpu-baseline --features 10000000 --factors 32 --nnz 16 2>/dev/null | tail -1
echo "# configs[2] shape on ONE GPU: 3.3e7 features, k=64, nnz=39, SGD (hogwild bench line)"
python bench.py --no-cpu-baseline --features 33000000 --factors 64 --nnz 39 2>/dev/null | tail -1
echo "# metric shape, exact MINIBATCH rule through the sharded driver (RCCL, 1 rank)"
python bench.py --no-cpu-baseline --force-sharded 2>/dev/null | tail -1
echo "# configs[3]: ALS k=64 (n=1e7, 16 nnz/row, 4.19 M rows)"
python scripts/gpu_als_probe.py 2>&1 | grep "^ALS"
echo "# configs[4] shape on ONE GPU: MCMC k=128, 1e8 features (and k=64, 32 nnz/row)"
python scripts/gpu_c5_probe.py 2>&1 | grep "^MCMC"
} > $OUT/named_configs.txt
cat $OUT/named_configs.txt | cut -c1-260
