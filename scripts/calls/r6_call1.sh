#!/bin/bash
# round 6, GPU call 1: where the 1.29 s of one-time slot preparation goes (FMX_TRACE_SETUP), and the headline's byte counters with the
# two streams ordered by events and the one-wavefront recurrence (FMX_HANDOFF=0 FMX_SCAN=serial: never tried together in round 5)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c1
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1"
T0=$(date +%s)
FMX_TRACE_SETUP=1 $B > $O/bench_trace.json 2> $O/bench_trace.err
echo "trace rc=$? $(( $(date +%s) - T0 )) s"; grep "fmx setup" $O/bench_trace.err | head -40; tail -c 600 $O/bench_trace.json
for c in FETCH_SIZE WRITE_SIZE; do
  T0=$(date +%s)
  FMX_HANDOFF=0 FMX_SCAN=serial timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- $B > $O/pmc_$c.out 2> $O/pmc_$c.err
  echo "events+serial $c rc=$? $(( $(date +%s) - T0 )) s"
done
python scripts/pmc_summary.py $O 8 > $O/pmc_summary.txt 2>&1
grep -A8 "^== pmc" $O/pmc_summary.txt | cut -c1-170
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
