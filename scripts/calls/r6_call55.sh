#!/bin/bash
# round 6, GPU call 55: the default bench with parity_vs_online MEASURED in the run (the device's reference-trajectory mode on the online side)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c55
mkdir -p $O
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6c55/bench_default.json"))
print(o["value"], o["roofline"]["frac"])
print(o.get("parity_vs_online_live"))
print({k: v for k, v in o["roofline"].items() if k.startswith("parity")})
print({k: o["parity_vs_online"].get(k) for k in ("rows", "pred_rms", "pred_mean_abs", "pred_max_abs", "w0_abs")})
PY
tail -3 $O/bench_default.err
