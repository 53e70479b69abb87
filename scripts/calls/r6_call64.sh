#!/bin/bash
# round 6, GPU call 64: the other bench modes and the bench tests on the final bench.py
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for m in hogwild minibatch; do
timeout 600 python bench.py --mode $m --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json; o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', o['value'], o['roofline'].get('frac'), o['roofline'].get('step_traffic_ratio'))"
done
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -2
