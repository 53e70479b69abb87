#!/bin/bash
# round 5, GPU call 6: k_apply_multi with its loads regrouped (entries + masks asked for with the sums; w written after the rows are asked for)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_sharded_driver.py -q -m gpu > $O/pytest_group.log 2>&1
tail -4 $O/pytest_group.log
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "shard" > $O/pytest_fuzz.log 2>&1
tail -3 $O/pytest_fuzz.log
LAG=2 timeout 600 python scripts/gpu_shard_probe.py 0 8,4 64 262144 2>&1 | grep "world=" | tee $O/shard_probe.txt
LAG=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o probe -- python scripts/gpu_shard_probe.py 0 8 64 262144 > /dev/null 2>&1
f=$(find $O/prof1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/shard_probe_p8_kernel_stats.csv; rm -rf $O/prof1
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r5c6/shard_probe_p8_kernel_stats.csv")))[:6]:
    print("  %-50s calls %5s avg %9.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
