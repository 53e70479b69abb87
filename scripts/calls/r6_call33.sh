#!/bin/bash
# round 6, GPU call 33: a run as ONE launch (k_run_fused): parity tests, rate, kernel stats; the three forms side by side
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c33
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conflict_free or long_runs" -x 2>&1 | tail -15
echo "one launch per run:"; timeout 300 python scripts/seq_rate.py 2>&1 | grep examples
echo "two launches per run:"; FMX_SEQ_RUNS_ONE=0 timeout 300 python scripts/seq_rate.py 2>&1 | grep examples
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o seq -- python scripts/seq_rate.py > $O/prof.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/seq_kernel_stats.csv \;
rm -rf $O/kt
grep -E "k_run_fused|k_run_apply|k_rowsums|k_scan|k_apply" $O/seq_kernel_stats.csv | cut -c1-60,150-400 | head
