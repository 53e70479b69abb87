#!/bin/bash
# round 5, GPU call 11: final confirmation -- smoke(), the whole GPU suite, the driver's default bench command (timed)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c11
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
grep -E "passed|failed" $O/pytest_all.log | tail -2
/usr/bin/time -v -o $O/bench_default_time.txt timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
grep -E "Elapsed|Maximum resident" $O/bench_default_time.txt
python -c "
import json; o=json.load(open('$O/bench_default.json')); print('default', o['value'], o['roofline']['frac'], o['config']['w0_chunk'], o['roofline'].get('per_config'), {k:v['per_rank_examples_per_s'] for k,v in o['shard_probe']['ranks'].items()}); print(o['parity_vs_online'])"
