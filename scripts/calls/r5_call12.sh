#!/bin/bash
# round 5, GPU call 12: the driver's default bench command on the final code, timed
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c12
mkdir -p $O
T0=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
T1=$(date +%s.%N)
echo "python bench.py: $(echo "$T1 - $T0" | bc) s wall on the box" | tee $O/bench_default_seconds.txt
python -c "
import json; o=json.load(open('$O/bench_default.json')); print('default', o['value'], o['roofline']['frac'], o['config']['w0_chunk'], o['roofline'].get('per_config'), {k:v['per_rank_examples_per_s'] for k,v in o['shard_probe']['ranks'].items()}); print(json.dumps(o['parity_vs_online'])[:1500])"
