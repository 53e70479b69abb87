#!/bin/bash
# round 6, GPU call 11: XCD-resident epoch vs two launches per batch at smaller explicit batches
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c11
mkdir -p $O
for bt in 64 128 256 512 1024; do
B="python bench.py --workload criteo --features 33000000 --nnz 39 --rows 262144 --batch $bt --no-extras --no-cpu-baseline --steps 3 --warmup 1"
FMX_XCD=0 timeout 300 $B > $O/two_$bt.json 2> $O/two_$bt.err
timeout 300 $B > $O/xcd_$bt.json 2> $O/xcd_$bt.err
python -c "
import json
a=json.load(open('$O/two_$bt.json')); b=json.load(open('$O/xcd_$bt.json'))
print('batch $bt: two launches %.1f M ex/s (%.2f us/batch), xcd %.1f M ex/s (%.2f us/batch)' % (a['value']/1e6, 1e3*a['ms_per_step']/(262144/$bt), b['value']/1e6, 1e3*b['ms_per_step']/(262144/$bt)))"
done
