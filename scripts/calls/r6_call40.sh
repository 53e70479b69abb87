#!/bin/bash
# round 6, GPU call 40: kernel durations of the one-launch batches (rocprofv3 --kernel-trace --stats of scripts/small_one_rate.py)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c40
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o so -- python scripts/small_one_rate.py > $O/prof.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/small_one_kernel_stats.csv \;
rm -rf $O/kt
grep launch $O/prof.log
grep -E "k_small_one|k_fused|k_apply_seg_scan" $O/small_one_kernel_stats.csv | cut -c1-50,330-460
