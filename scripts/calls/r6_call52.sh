#!/bin/bash
# round 6, GPU call 52 (as calls 24, 32 and 44, on the final commit): what the driver runs at round end -- smoke(), the whole -m gpu suite, the default bench -- plus the kernel trace of the
# headline (rocprofv3 --kernel-trace --stats) whose average k_fused duration the bench's HIP-event time must agree with
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c52
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -4 $O/pytest_gpu.txt | head -2
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6c52/bench_default.json"))
print(o["value"], o["ms_per_step"], {k: v for k, v in o["roofline"].items() if not isinstance(v, (dict, list, str))})
print({k: (o[k].get("value") if isinstance(o.get(k), dict) else None) for k in ("c2", "criteo", "criteo_8shard", "als", "mcmc", "mcmc_c5")})
print(o.get("cpu_baseline"))
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_under_rocprof.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf $O/kt
