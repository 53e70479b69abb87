#!/bin/bash
# round 6, GPU call 3: which allocation of the slot preparation is still slow (FMX_TRACE_ALLOC), and the new safety/adapter tests
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c3
mkdir -p $O
FMX_TRACE_ALLOC=1 FMX_TRACE_SETUP=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_trace.json 2> $O/bench_trace.err; echo "bench rc=$?"
grep "fmx setup\|fmx alloc" $O/bench_trace.err | head -60
timeout 900 python -m pytest tests/test_gpu_safety.py tests/test_gpu_adapter.py -x -q -m gpu > $O/pytest_new.txt 2>&1; echo "new tests rc=$?"; tail -25 $O/pytest_new.txt
