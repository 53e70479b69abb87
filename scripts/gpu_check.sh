#!/bin/bash
# one gpurun call: gpu tests + smoke + bench variants (+ optional rocprof). Outputs under gpurun_out/.
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1) > gpurun_out/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/pytest.log | head -40
grep -E "Mismatched|Max absolute|Max relative|^E  +assert|FmxError" gpurun_out/pytest.log | head -60
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8) > gpurun_out/smoke.log 2>&1
cat gpurun_out/smoke.log
