#!/bin/bash
# round 5, GPU call 9: smoke() with its new recurrence leg, the tightened online band
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c9
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "online_loop" -s 2>&1 | grep -E "parity_vs_online|passed|failed|Error" | tee $O/online_band.txt
