#!/bin/bash
# the placement of the parameter tables, process by process: first fit (--place 1) against the library's default (an arena of chunks from
# two memory classes) -- the default bench workload, no CPU legs.  usage: gpurun -- 'bash scripts/gpu_place_ab.sh [rounds=3]'
R=${1:-3}
mkdir -p gpurun_out
out=gpurun_out/place_ab.txt
: > $out
timeout 600 python -m pytest tests/test_gpu_zz_placement.py -x -q 2>&1 | tail -5 | tee -a $out
for i in $(seq 1 $R); do
  for p in 1 0; do
    echo "== --place $p (process $i)" >> $out
    timeout 300 python bench.py --place $p --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>&1 | python3 -c '
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l)
        print("  %.1f M ex/s  %.3f ms/step  frac %.4f  placement %s" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"], json.dumps(d["config"].get("placement"))))
    elif l: print("  " + l)
' >> $out
  done
done
cat $out
