#!/bin/bash
# round 6, GPU call 10: XCD-resident epoch, which touches matter (FMX_XCD_FLAGS)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c10
mkdir -p $O
B="python bench.py --workload criteo --features 33000000 --nnz 39 --rows 1048576 --no-extras --no-cpu-baseline --steps 3 --warmup 1"
for fl in 0 1 2 3; do
rm -f $O/hops_$fl.txt
FMX_XCD_FLAGS=$fl FMX_XCD_TRACE=$O/hops_$fl.txt timeout 300 $B > $O/criteo_xcd_$fl.json 2> $O/criteo_xcd_$fl.err; echo "flags $fl rc=$?"
python -c "
import json; o=json.load(open('$O/criteo_xcd_$fl.json')); print(o['value'], o['ms_per_step'])"
sed -n 45,50p $O/hops_$fl.txt
done
