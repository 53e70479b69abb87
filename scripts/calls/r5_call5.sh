#!/bin/bash
# round 5, GPU call 5: the software-pipelined short-row kernels -- parity (group / sharded tests), then the P = 8 probe by launch size
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_sharded_driver.py tests/test_gpu_bench.py -q -m gpu > $O/pytest_group.log 2>&1
tail -5 $O/pytest_group.log
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "large_batches or shard" > $O/pytest_fuzz.log 2>&1
tail -3 $O/pytest_fuzz.log
for over in 1 2 4 65536; do FMX_MULTI_OVER=$over LAG=2 timeout 600 python scripts/gpu_shard_probe.py 0 8,4 64 262144 2>&1 | grep "world=" | sed "s/^/over$over /"; done | tee $O/shard_probe_over.txt
FMX_MULTI_OVER=1 LAG=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o probe -- python scripts/gpu_shard_probe.py 0 8 64 262144 > /dev/null 2>&1
f=$(find $O/prof1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/shard_probe_p8_over1_kernel_stats.csv; rm -rf $O/prof1
FMX_MULTI_OVER=2 LAG=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o probe -- python scripts/gpu_shard_probe.py 0 8 64 262144 > /dev/null 2>&1
f=$(find $O/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/shard_probe_p8_over2_kernel_stats.csv; rm -rf $O/prof2
python - <<'PY'
import csv
for f in ("gpurun_out/r5c5/shard_probe_p8_over1_kernel_stats.csv", "gpurun_out/r5c5/shard_probe_p8_over2_kernel_stats.csv"):
    print(f)
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  %-50s calls %5s avg %9.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
