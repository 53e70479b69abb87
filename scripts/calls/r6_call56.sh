#!/bin/bash
# round 6, GPU call 56: the bench tests and the default bench on the final bench.py
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c56
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -2
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6c56/bench_default.json"))
print(o["value"], o["roofline"]["frac"], o["parity_vs_online_live"]["pred_mean_abs"], o["parity_vs_online"]["source"][:90])
PY
