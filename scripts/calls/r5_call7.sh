#!/bin/bash
# round 5, GPU call 7: the evidence run -- whole GPU suite, the driver's default bench line, kernel stats + same-box counters of the headline,
# request counters of the P = 8 shard probe, the cost of bias lag 1
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c7
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
tail -6 $O/pytest_all.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; o=json.load(open('$O/bench_default.json')); print('default', o['value'], o['roofline']['frac'], o['config']['w0_chunk'], o['roofline'].get('per_config'), {k:v['per_rank_examples_per_s'] for k,v in o['shard_probe']['ranks'].items()})"
B="python bench.py --no-extras --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o bench -- $B --steps 5 --warmup 1 > $O/bench_under_rocprof.json 2>/dev/null
f=$(find $O/bench_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv; rm -rf $O/bench_trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- $B --steps 3 --warmup 1 > /dev/null 2>&1
done
for c in TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
  LAG=2 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_probe_$c -o probe -- python scripts/gpu_shard_probe.py 0 8 64 262144 > /dev/null 2>&1
done
python scripts/pmc_summary.py $O 6 > $O/pmc_summary.txt 2>&1
grep -A6 "^== pmc" $O/pmc_summary.txt | cut -c1-170
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_probe_TCC_EA0_RDREQ_sum $O/pmc_probe_TCC_EA0_WRREQ_sum
for lag in 1 2 3; do timeout 300 $B --steps 10 --warmup 2 --bias-lag $lag > $O/bench_lag$lag.json 2>> $O/bench.err; python -c "
import json; o=json.load(open('$O/bench_lag$lag.json')); print('lag$lag', o['value'], o['ms_per_step'], o['roofline']['frac'])"; done
head -6 $O/bench_kernel_stats.csv | cut -c1-70,200-330
