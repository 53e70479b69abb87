#!/bin/bash
# two feature shards on one device, plain allocations vs arenas, process by process
mkdir -p gpurun_out
out=gpurun_out/place_ab2.txt
: > $out
for i in 1 2 3; do
  for p in 1 0; do
    echo "== --gpus 2 --same-device --place $p (process $i)" >> $out
    timeout 300 python bench.py --gpus 2 --same-device --no-cpu-baseline --place $p 2>/dev/null | python3 -c '
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l)
        print("  %.1f M ex/s  %.3f ms/step  phases %s  placement %s" % (d["value"] / 1e6, d["ms_per_step"], json.dumps(d["phases_ms_per_batch"]), json.dumps(d["config"].get("placement"))))
' >> $out
  done
done
cat $out
