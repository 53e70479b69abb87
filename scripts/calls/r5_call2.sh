#!/bin/bash
# round 5, GPU call 2: the parallel-in-time bias recurrence (k_scan_pit) and the short-row shard kernels (k_rowsums_multi / k_apply_multi)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiled_recurrence or side_stream or many_batches or kilo or newton or default_bias or fused_minibatch_matches or deterministic" > $O/pytest_parity.log 2>&1
tail -5 $O/pytest_parity.log
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_sharded_driver.py -x -q -m gpu > $O/pytest_group.log 2>&1
tail -5 $O/pytest_group.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_stability.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_full.log 2>&1
tail -5 $O/pytest_full.log
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2"
timeout 300 $B > $O/bench_pit32.json 2> $O/bench.err
FMX_SCAN=serial timeout 300 $B > $O/bench_serial32.json 2>> $O/bench.err
timeout 300 $B --w0-chunk 1 > $O/bench_pit1.json 2>> $O/bench.err
timeout 300 $B --w0-chunk 256 > $O/bench_pit256.json 2>> $O/bench.err
for f in pit32 serial32 pit1 pit256; do python -c "
import json; o=json.load(open('$O/bench_$f.json')); print('$f', o['value'], o['ms_per_step'], o['roofline']['frac'])"; done
timeout 600 python scripts/gpu_shard_probe.py 32 8,4,2 64 262144 > $O/shard_probe.txt 2>&1
cat $O/shard_probe.txt | grep -v amdgpu.ids
FMX_SCAN=serial timeout 600 python scripts/gpu_shard_probe.py 256 8 64 262144 > $O/shard_probe_serial256.txt 2>&1
cat $O/shard_probe_serial256.txt | grep -v amdgpu.ids
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe8 -o probe -- python scripts/gpu_shard_probe.py 32 8 64 262144 > /dev/null 2>&1
for d in prof_bench prof_probe8; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
head -12 $O/prof_bench_kernel_stats.csv | cut -c1-200
head -14 $O/prof_probe8_kernel_stats.csv | cut -c1-200
