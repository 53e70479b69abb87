#!/bin/bash
# round 6, GPU call 19: the deferred pass by segments in flight per wavefront (U = 4 / 8 / 16)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c19
mkdir -p $O
for u in 8 16 4 8 16; do
FMX_SEG_U=$u timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $O/u_$u.json 2> $O/u_$u.err
python -c "
import json
a=json.load(open('$O/u_$u.json')); print('U=$u: %.1f M ex/s, %.3f ms/step, frac %.4f' % (a['value']/1e6, a['ms_per_step'], a['roofline']['frac']))"
done
