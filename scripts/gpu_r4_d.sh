#!/bin/bash
# round 4, fourth GPU call: suite, 32-byte deferred records (k_apply_seg time in the kernel stats), Criteo rate, default bench
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- $B > $OUT/bench_under_rocprof.json 2>/dev/null
cp $OUT/bench_trace/*/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null || cp $OUT/bench_trace/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv
grep "k_fused\|k_apply_seg" $OUT/bench_kernel_stats.csv | cut -c1-40,180-330
cut -c1-300 $OUT/bench_under_rocprof.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/criteo_trace -o criteo -- $B --workload criteo --rows 1048576 --steps 3 --warmup 1 > $OUT/criteo_under_rocprof.json 2>/dev/null
cp $OUT/criteo_trace/*/criteo_kernel_stats.csv $OUT/criteo_kernel_stats.csv 2>/dev/null || cp $OUT/criteo_trace/criteo_kernel_stats.csv $OUT/criteo_kernel_stats.csv
grep "k_fused\|k_apply_seg" $OUT/criteo_kernel_stats.csv | cut -c1-40,180-330
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload criteo --rows 1048576 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" | cut -c1-160
timeout 900 python bench.py 2>$OUT/bench_default.err | grep "^{" > $OUT/bench_default.json
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4d/bench_default.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "ms", d["ms_per_step"])
for k in ("predict","c2","criteo","mcmc_c5","als","mcmc"):
    v=d.get(k,{}); print(k, v.get("value"), v.get("ms_per_step"), v.get("error"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"))
PY
find $OUT -name "*.csv" -size +3M -delete; rm -rf $OUT/bench_trace $OUT/criteo_trace
