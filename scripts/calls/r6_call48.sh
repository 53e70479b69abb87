#!/bin/bash
# round 6, GPU call 48: the default bench once more on the final bench.py (new flat scalars), and the bench tests
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c48
mkdir -p $O
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6c48/bench_default.json"))
print(o["value"], o["ms_per_step"], {k: v for k, v in o["roofline"].items() if not isinstance(v, (dict, list, str))})
print(o["criteo"]["roofline"]["kernel"])
PY
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -2
