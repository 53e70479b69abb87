#!/bin/bash
# round 6, GPU call 28: the whole suite under glibc's heap checks (MALLOC_CHECK_=3, MALLOC_PERTURB_): is the one-off abort of call 25 a host heap overrun?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c28
mkdir -p $O
MALLOC_CHECK_=3 MALLOC_PERTURB_=165 timeout 2400 python -m pytest tests -q -m gpu --capture=no -p no:faulthandler > $O/suite.txt 2>&1; rc=$?
echo "suite rc=$rc $(grep -E '[0-9]+ passed' $O/suite.txt | tail -1)"
grep -n -i "malloc\|free()\|corrupt\|invalid pointer\|double free" $O/suite.txt | head -10
if [ $rc -ne 0 ]; then tail -c 2500 $O/suite.txt; fi
