#!/bin/bash
# round 6, GPU call 47: the whole -m gpu suite three more times on the final code (is anything flaky? the one-off abort of call 25?)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c47
mkdir -p $O
for i in 1 2 3; do
timeout 1500 python -m pytest tests -q -m gpu --capture=no -p no:faulthandler > $O/suite_$i.txt 2>&1; rc=$?
echo "suite $i rc=$rc $(grep -E '[0-9]+ passed' $O/suite_$i.txt | tail -1)"
if [ $rc -ne 0 ]; then tail -c 3000 $O/suite_$i.txt; break; fi
rm -f $O/suite_$i.txt
done
