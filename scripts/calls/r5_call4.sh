#!/bin/bash
# round 5, GPU call 4: k_scan_pit v2 (chunk maps kept in LDS, DPP sums / scans) -- parity first, then the whole suite, the shard probe (lag 1 / 2,
# fill of a round), the bench quick lines
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tiled_recurrence or side_stream or newton or kilo" > $O/pytest_pit.log 2>&1
tail -4 $O/pytest_pit.log
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
tail -8 $O/pytest_all.log
for lag in 1 2; do LAG=$lag timeout 600 python scripts/gpu_shard_probe.py 0 8 64 262144 2>&1 | grep "world=" | sed "s/^/lag$lag /" ; done | tee $O/shard_probe.txt
for fill in 16 20 28 32; do FMX_MULTI_FILL=$fill LAG=2 timeout 600 python scripts/gpu_shard_probe.py 0 8 64 262144 2>&1 | grep "world=" | sed "s/^/fill$fill /"; done | tee -a $O/shard_probe.txt
LAG=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe8 -o probe -- python scripts/gpu_shard_probe.py 0 8 64 262144 > /dev/null 2>&1
f=$(find $O/prof_probe8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/shard_probe_p8_lag1_kernel_stats.csv; rm -rf $O/prof_probe8
head -6 $O/shard_probe_p8_lag1_kernel_stats.csv | cut -c1-60,200-330
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2"
timeout 300 $B > $O/bench_pit32.json 2> $O/bench.err
timeout 300 $B --w0-chunk 1 > $O/bench_pit1.json 2>> $O/bench.err
for f in pit32 pit1; do python -c "
import json; o=json.load(open('$O/bench_$f.json')); print('$f', o['value'], o['ms_per_step'], o['roofline']['frac'])"; done
