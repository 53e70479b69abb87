#!/bin/bash
# round 6, GPU call 27: the whole suite without output capture, repeated, to see what the runtime says before the abort of call 25
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c27
mkdir -p $O
for i in 1 2 3 4; do
timeout 1500 python -m pytest tests -q -m gpu --capture=no -p no:faulthandler > $O/suite_$i.txt 2>&1; rc=$?
echo "suite $i rc=$rc $(grep -E '[0-9]+ passed' $O/suite_$i.txt | tail -1)"
if [ $rc -ne 0 ]; then tail -c 3000 $O/suite_$i.txt; break; fi
rm -f $O/suite_$i.txt
done
