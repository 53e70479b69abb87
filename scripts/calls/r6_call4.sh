#!/bin/bash
# round 6, GPU call 4: why a 300 000-row batch does not take k_scan_pit (FMX_TRACE_PIT); the slot preparation after the CPU baseline leg
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c4
mkdir -p $O
FMX_TRACE_PIT=1 timeout 600 python -m pytest tests/test_gpu_safety.py -x -q -m gpu -k "pieces" > $O/pytest_pieces.txt 2>&1; echo "pieces rc=$?"; grep "fmx pit" $O/pytest_pieces.txt | sort | uniq -c | head; tail -5 $O/pytest_pieces.txt
FMX_TRACE_ALLOC=1 FMX_TRACE_SETUP=1 timeout 600 python bench.py --no-extras --steps 3 --warmup 1 > $O/bench_trace.json 2> $O/bench_trace.err; echo "bench rc=$?"
grep "fmx setup\|fmx alloc" $O/bench_trace.err | head -60
