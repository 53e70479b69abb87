#!/bin/bash
# round 4, final evidence call: the whole GPU suite, the default bench (timed), kernel stats of the headline / ALS / Criteo / configs[4] runs
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > $OUT/pytest_gpu.log 2>&1
tail -2 $OUT/pytest_gpu.log
T0=$(date +%s)
timeout 1500 python bench.py 2>$OUT/bench_default.err | grep "^{" > $OUT/bench_default.json
echo "bench.py default run: $(( $(date +%s) - T0 )) s" | tee $OUT/bench_default_seconds.txt
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4e/bench_default.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "ms", d["ms_per_step"], d["config"]["placement"])
for k in ("predict","c2","criteo","mcmc_c5","als","mcmc","hogwild","minibatch_two_pass"):
    v=d.get(k,{}); print(k, v.get("value"), v.get("ms_per_step"), v.get("error"), (v.get("roofline") or {}).get("frac"))
PY
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/als_trace -o als -- $B --method als --steps 2 --warmup 1 > $OUT/als_under_rocprof.json 2>/dev/null
cp $OUT/als_trace/*/als_kernel_stats.csv $OUT/als_kernel_stats.csv 2>/dev/null || cp $OUT/als_trace/als_kernel_stats.csv $OUT/als_kernel_stats.csv
grep "k_als_draw\|k_als_rows" $OUT/als_kernel_stats.csv | cut -c1-45,150-260
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace -o c5 -- $B --method mcmc --factors 128 --features 100000000 --steps 2 --warmup 1 > $OUT/c5_under_rocprof.json 2>/dev/null
cp $OUT/c5_trace/*/c5_kernel_stats.csv $OUT/mcmc_c5_kernel_stats.csv 2>/dev/null || cp $OUT/c5_trace/c5_kernel_stats.csv $OUT/mcmc_c5_kernel_stats.csv
grep "k_als_draw\|k_als_rows\|unseen" $OUT/mcmc_c5_kernel_stats.csv | cut -c1-45,150-260
cd $GRAFT_REPO_ROOT
find $OUT -name "*.csv" -size +3M -delete; rm -rf $OUT/als_trace $OUT/c5_trace
