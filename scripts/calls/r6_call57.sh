#!/bin/bash
# round 6, GPU call 57: the default bench with roofline.traffic MEASURED in the run (two rocprofv3 --pmc passes in child processes)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c57
mkdir -p $O
T0=$(date +%s)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6c57/bench_default.json"))
r = o["roofline"]
print(o["value"], r["frac"], r["traffic"], r.get("step_traffic_ratio"))
print(r["traffic_source"][:300])
print(r.get("step_traffic_ratio_source", "")[:200])
PY
tail -3 $O/bench_default.err
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -2
