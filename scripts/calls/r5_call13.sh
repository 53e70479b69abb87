#!/bin/bash
# round 5, GPU call 13: same-box byte counters of the headline -- with the one-wavefront recurrence (FMX_SCAN=serial), which call 8's passes
# suggest the profiler's kernel serialisation tolerates, and once more with the default to see which kernel stalls
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c13
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  T0=$(date +%s)
  FMX_SCAN=serial timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- $B > $O/pmc_$c.out 2> $O/pmc_$c.err
  echo "serial $c rc=$? $(( $(date +%s) - T0 )) s"
done
T0=$(date +%s)
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcpit_FETCH_SIZE -o bench -- $B > $O/pmcpit.out 2> $O/pmcpit.err
echo "pit FETCH_SIZE rc=$? $(( $(date +%s) - T0 )) s"
python scripts/pmc_summary.py $O 6 > $O/pmc_summary.txt 2>&1
grep -A6 "^== pmc" $O/pmc_summary.txt | cut -c1-170
grep -h "value" $O/pmc_FETCH_SIZE.out | cut -c1-200
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmcpit_FETCH_SIZE
