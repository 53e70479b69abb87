#!/bin/bash
# round 5, GPU call 3: the whole GPU suite, the default bench line (incl. shard_probe), the P = 8 shard probe under rocprof, request counters at the
# configs[4] shape
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c3
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
tail -6 $O/pytest_all.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
o = json.load(open("gpurun_out/r5c3/bench_default.json"))
print("value", o["value"], "frac", o["roofline"]["frac"], "w0_chunk", o["config"]["w0_chunk"])
print("roofline keys", {k: o["roofline"].get(k) for k in ("predict", "step_traffic_ratio", "per_config", "shard_probe")})
print("shard_probe", json.dumps(o.get("shard_probe"))[:1500])
for k in ("c2", "criteo", "als", "mcmc", "mcmc_c5"):
    print(k, (o.get(k) or {}).get("value"), (o.get(k) or {}).get("error"))
PY
LAG=2 timeout 600 python scripts/gpu_shard_probe.py 0 8,4,2 64 262144 > $O/shard_probe_lag2.txt 2>&1
LAG=1 timeout 600 python scripts/gpu_shard_probe.py 0 8 64 262144 > $O/shard_probe_lag1.txt 2>&1
grep -h "world=" $O/shard_probe_lag2.txt $O/shard_probe_lag1.txt
LAG=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe8 -o probe -- python scripts/gpu_shard_probe.py 0 8 64 262144 > /dev/null 2>&1
f=$(find $O/prof_probe8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/shard_probe_p8_kernel_stats.csv; rm -rf $O/prof_probe8
head -8 $O/shard_probe_p8_kernel_stats.csv | cut -c1-60,200-330
C5="python bench.py --method mcmc --features 100000000 --factors 128 --nnz 16 --steps 2 --warmup 1 --no-cpu-baseline"
for c in TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_c5_$c -o c5 -- $C5 > /dev/null 2> $O/pmc_c5_$c.err
done
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- $C5 > $O/c5_under_rocprof.json 2>/dev/null
f=$(find $O/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv; rm -rf $O/prof_c5
python scripts/pmc_summary.py $O 10 > $O/pmc_c5_summary.txt 2>&1
grep -A10 "^== pmc" $O/pmc_c5_summary.txt | cut -c1-170
rm -rf $O/pmc_c5_TCC_EA0_RDREQ_sum $O/pmc_c5_TCC_EA0_WRREQ_sum
head -8 $O/c5_kernel_stats.csv | cut -c1-70,200-330
