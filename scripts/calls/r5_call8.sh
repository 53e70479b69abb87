#!/bin/bash
# round 5, GPU call 8: fuzz soak over seeds the suite does not use (the new recurrence / short-row kernels), same-box byte counters of the headline
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c8
mkdir -p $O
bash scripts/gpu_fuzz_soak.sh 1000:1030 900
cp gpurun_out/fuzz_soak.txt $O/fuzz_soak.txt
B="python bench.py --no-extras --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- $B --steps 3 --warmup 1 > $O/pmc_$c.out 2> $O/pmc_$c.err
  tail -3 $O/pmc_$c.err
done
python scripts/pmc_summary.py $O 6 > $O/pmc_summary.txt 2>&1
grep -A6 "^== pmc" $O/pmc_summary.txt | cut -c1-170
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
