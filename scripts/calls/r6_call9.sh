#!/bin/bash
# round 6, GPU call 9: XCD-resident epoch, tests + hop trace
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_xcd.py -x -q -m gpu > $O/pytest_xcd.txt 2>&1; echo "xcd tests rc=$?"; tail -12 $O/pytest_xcd.txt
B="python bench.py --workload criteo --features 33000000 --nnz 39 --rows 1048576 --no-extras --no-cpu-baseline --steps 3 --warmup 1"
rm -f $O/hops.txt
FMX_XCD_TRACE=$O/hops.txt timeout 300 $B > $O/criteo_xcd.json 2> $O/criteo_xcd.err; echo "xcd rc=$?"
python -c "
import json; o=json.load(open('$O/criteo_xcd.json')); print(o['value'], o['ms_per_step'])"
head -3 $O/hops.txt; sed -n 40,52p $O/hops.txt
