#!/bin/bash
# bench variants, one JSON line each -> gpurun_out/bench_variants.log
mkdir -p gpurun_out
: > gpurun_out/bench_variants.log
run() { (timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>&1 | grep '^{' ) >> gpurun_out/bench_variants.log; }
run --mode hogwild --apply store
run --mode hogwild --apply atomic
run --mode minibatch --apply segmented
run --mode minibatch --apply store
run --mode minibatch --apply segmented --batch 8192
run --mode minibatch --apply segmented --batch 65536
run --mode minibatch --apply segmented --no-bias-lag
run --mode minibatch --apply segmented --batch 65536 --no-bias-lag
python - <<'PY'
import json
for line in open('gpurun_out/bench_variants.log'):
    d=json.loads(line); r=d['roofline'] or {}
    print("%-9s %-9s B=%-7s  %8.1f Mex/s  %7.2f ms/step | %-11s %7.1f GB/s frac %.3f avg %.4f ms" % (d['config']['mode'], d['config']['apply'], d['config']['batch'], d['value']/1e6, d['ms_per_step'], r.get('kernel'), r.get('achieved',0), r.get('frac',0), r.get('avg_launch_ms',0)))
PY
