#!/bin/bash
# round 6, GPU call 58: the default bench (traffic measured in the run) and the same command under rocprofv3 --kernel-trace --stats (the nested counter passes must be skipped)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c58
mkdir -p $O
T0=$(date +%s)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - T0 )) s"
python -c "
import json; r = json.load(open('gpurun_out/r6c58/bench_default.json'))['roofline']; print(r['frac'], r['traffic'], r['step_traffic_ratio'], r['traffic_source'][:40])"
T0=$(date +%s)
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- python bench.py > $O/bench_profiled.json 2> $O/bench_profiled.err; echo "profiled default bench rc=$? $(( $(date +%s) - T0 )) s"
rm -rf $O/kt
python -c "
import json; r = json.load(open('gpurun_out/r6c58/bench_profiled.json'))['roofline']; print(r['frac'], r['traffic'], r['traffic_source'][:40])"
