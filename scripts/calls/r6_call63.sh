#!/bin/bash
# round 6, GPU call 63: smoke() and the whole -m gpu suite on the final commit
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c63
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $O/pytest_gpu.txt | tail -1
