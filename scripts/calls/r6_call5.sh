#!/bin/bash
# round 6, GPU call 5: the whole -m gpu suite on the round-6 code; the slot preparation right after it (call 2 saw 1.0 s there, calls 3/4 0.04 s on a
# quiet box); the default bench
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c5
mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -8 $O/pytest_gpu.txt
for i in 1 2; do
FMX_TRACE_ALLOC=1 FMX_TRACE_SETUP=1 timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_noextras_$i.json 2> $O/bench_noextras_$i.err; echo "bench $i rc=$?"
grep "fmx setup\|fmx alloc" $O/bench_noextras_$i.err | head -20
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6c5/bench_default.json"))
print(o["value"], o["ms_per_step"], {k: v for k, v in o["roofline"].items() if not isinstance(v, (dict, list, str))})
PY
