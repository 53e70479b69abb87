#!/bin/bash
# round 6, GPU call 41: device time stamps of a one-launch batch (FMX_SMALL_TRACE)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c41
mkdir -p $O
FMX_SMALL_TRACE=$O/trace.txt timeout 300 python scripts/small_one_rate.py 2>&1 | grep launch
cat $O/trace.txt
