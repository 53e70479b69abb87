#!/bin/bash
# round 6, GPU call 12: the headline step by batch size (the deferred pass shrinks with the batch: collisions per example ~ batch / n)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c12
mkdir -p $O
for bt in 262144 196608 131072 98304 65536; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --batch $bt > $O/b_$bt.json 2> $O/b_$bt.err
python -c "
import json
a=json.load(open('$O/b_$bt.json')); r=a['roofline']
print('batch $bt: %.1f M ex/s, %.3f ms/step, frac %.4f, deferred/example %.3f, launches %s' % (a['value']/1e6, a['ms_per_step'], r['frac'], r.get('deferred_features_per_example', 0), r.get('launches')))"
done
