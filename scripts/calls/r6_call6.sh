#!/bin/bash
# round 6, GPU call 6: byte counters of the headline with NO environment knobs (round-5 verdict item 5: the handle's probe finds the serialised
# dispatch of a counter-collecting profiler and orders its streams with events); kernel trace of the same command
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c6
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  T0=$(date +%s)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- $B > $O/pmc_$c.out 2> $O/pmc_$c.err
  echo "no knobs $c rc=$? $(( $(date +%s) - T0 )) s"; tail -c 400 $O/pmc_$c.out | grep -o '"status[^,]*,' | head -2
done
python scripts/pmc_summary.py $O 8 > $O/pmc_summary.txt 2>&1
grep -A8 "^== pmc" $O/pmc_summary.txt | cut -c1-170
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
T0=$(date +%s)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > $O/kt.out 2> $O/kt.err
echo "kernel trace rc=$? $(( $(date +%s) - T0 )) s"
cp $O/kt/*/bench_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -12 $O/bench_kernel_stats.csv | cut -c1-200
rm -rf $O/kt
