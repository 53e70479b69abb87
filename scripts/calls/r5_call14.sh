#!/bin/bash
# round 5, GPU call 14: the byte counters once more, exactly as round 4 took them (from /tmp, absolute path of bench.py)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5c14
mkdir -p $O
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
for c in FETCH_SIZE WRITE_SIZE; do
  T0=$(date +%s)
  timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- $B --steps 3 --warmup 1 > $O/pmc_$c.out 2> $O/pmc_$c.err
  rc=$?
  echo "$c rc=$rc $(( $(date +%s) - T0 )) s"
  [ $rc -ne 0 ] && break
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O 6 > $O/pmc_summary.txt 2>&1
grep -A6 "^== pmc" $O/pmc_summary.txt | cut -c1-170
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
