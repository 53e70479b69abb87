#!/bin/bash
# round 6, GPU call 65: the default bench with the order-noise yardstick measured in the run (the reference's trajectory against itself, rows shuffled inside windows)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c65
mkdir -p $O
T0=$(date +%s)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - T0 )) s"
python -c "
import json; o = json.load(open('gpurun_out/r6c65/bench_default.json')); r = o['roofline']
print(o['value'], r['frac'], r['traffic'], r['step_traffic_ratio'])
print(o['parity_vs_online_live'])
print({k: v for k, v in r.items() if k.startswith(('parity', 'order'))})"
tail -2 $O/bench_default.err
