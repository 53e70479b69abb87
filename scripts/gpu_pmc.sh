#!/bin/bash
# PMC passes (separate, as MI355X_MICROARCH.md prescribes): FETCH_SIZE and WRITE_SIZE for the bench kernels + a
# calibration gather of known size (predict with k1=0: rows*nnz*256 B of V + 256 B of entries per row).
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/hog_$c -o hog -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --mode hogwild > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/mb_$c -o mb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --mode minibatch --rows 1048576 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -o cal -- python $GRAFT_REPO_ROOT/scripts/gpu_calib.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT
