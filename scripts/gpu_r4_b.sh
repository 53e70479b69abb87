#!/bin/bash
# round 4, second GPU call: the GPU suite (side stream, arena cache, 8 shards as typed), side-stream A/B, w-store variant, default bench,
# same-box PMC of the headline's two kernels, kernel stats of the Criteo-shaped run
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -30 ) > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
( timeout 300 python scripts/gpu_ab_wside.py ) > $OUT/ab_wside.txt 2>&1; cat $OUT/ab_wside.txt
( timeout 400 python scripts/gpu_ab_variants.py wst 2 fused ) > $OUT/ab_wst.txt 2>&1; tail -3 $OUT/ab_wst.txt
timeout 900 python bench.py 2>$OUT/bench_default.err | grep "^{" > $OUT/bench_default.json
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4b/bench_default.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "placement", d["config"]["placement"])
print("predict", json.dumps(d.get("predict")))
for k in ("c2","criteo","mcmc_c5","als","mcmc"):
    v=d.get(k,{}); print(k, v.get("value"), v.get("ms_per_step"), v.get("error"), (v.get("config") or {}).get("placement"))
PY
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o bench -- $B --steps 3 --warmup 1 > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/criteo_trace -o criteo -- $B --workload criteo --rows 1048576 --steps 3 --warmup 1 > $OUT/criteo_under_rocprof.json 2>/dev/null
cp $OUT/criteo_trace/*/criteo_kernel_stats.csv $OUT/criteo_kernel_stats.csv 2>/dev/null || cp $OUT/criteo_trace/criteo_kernel_stats.csv $OUT/criteo_kernel_stats.csv
head -4 $OUT/criteo_kernel_stats.csv | cut -c1-60,150-260
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT 6 > $OUT/pmc_summary.txt 2>&1
grep -A5 "^== pmc" $OUT/pmc_summary.txt | cut -c1-160
find $OUT -name "*.csv" -size +3M -delete
rm -rf $OUT/criteo_trace
du -sh $OUT
