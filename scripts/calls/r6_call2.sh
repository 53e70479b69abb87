#!/bin/bash
# round 6, GPU call 2: first run of the round-6 changes (allocator through the virtual-memory API, safe hand-offs, patched reference driver):
# the new tests first, then the whole -m gpu suite, then the default bench with the setup trace
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_safety.py tests/test_gpu_adapter.py -x -q -m gpu > $O/pytest_new.txt 2>&1; echo "new tests rc=$?"; tail -15 $O/pytest_new.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -8 $O/pytest_gpu.txt
FMX_TRACE_SETUP=1 timeout 600 python bench.py --no-extras --steps 20 --warmup 3 > $O/bench_noextras.json 2> $O/bench_noextras.err; echo "bench rc=$?"
grep "fmx setup" $O/bench_noextras.err | head -20
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6c2/bench_noextras.json"))
print(o["value"], o["ms_per_step"], {k: v for k, v in o["roofline"].items() if not isinstance(v, (dict, list))})
PY
