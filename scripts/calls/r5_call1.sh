#!/bin/bash
# round 5, GPU call 1: the sub-piece bias recurrence (k_scan1<.., 16/32/64/128>, scan_small below 64) against the oracle, then the headline
# with the new default micro-chunk and the recurrence kernel's time per batch
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r5c1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiled_recurrence or side_stream or many_batches or kilo or default_bias or fused_minibatch_matches" > gpurun_out/r5c1/pytest_parity.log 2>&1
tail -5 gpurun_out/r5c1/pytest_parity.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_stability.py -x -q -m gpu > gpurun_out/r5c1/pytest_full.log 2>&1
tail -5 gpurun_out/r5c1/pytest_full.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r5c1/bench_chunk32.json 2> gpurun_out/r5c1/bench_chunk32.err
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --w0-chunk 64 > gpurun_out/r5c1/bench_chunk64.json 2>> gpurun_out/r5c1/bench_chunk32.err
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --w0-chunk 256 > gpurun_out/r5c1/bench_chunk256.json 2>> gpurun_out/r5c1/bench_chunk32.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r5c1/prof32" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-extras --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r5c1/prof64" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-extras --no-cpu-baseline --steps 5 --warmup 1 --w0-chunk 64 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for d in prof32 prof64; do f=$(find gpurun_out/r5c1/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > gpurun_out/r5c1/${d}_kernel_stats_head.csv; find gpurun_out/r5c1/$d -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r5c1/$d -name "*.db" -delete; done
cat gpurun_out/r5c1/bench_chunk32.json | cut -c1-600
cat gpurun_out/r5c1/prof32_kernel_stats_head.csv
