#!/bin/bash
# round 6, GPU call 20: the opt-in paths under the tests of the default ones: FMX_XCD=1 (XCD-resident epoch wherever it is eligible),
# FMX_ROW_STRIDE_KP=1 (rows padded to the power of two, the round-5 layout), FMX_GROUP_THREADS=1
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c20
mkdir -p $O
FMX_XCD=1 timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_stability.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -k "not xcd" > $O/xcd.txt 2>&1; echo "xcd=1 rc=$?"; grep -E "passed|failed" $O/xcd.txt | tail -2
FMX_ROW_STRIDE_KP=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_als.py tests/test_gpu_group.py -q -m gpu -k "not odd_k" > $O/kp.txt 2>&1; echo "stride kp rc=$?"; grep -E "passed|failed" $O/kp.txt | tail -2
FMX_GROUP_THREADS=1 timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_configs.py -q -m gpu -k "not schedules" > $O/threads.txt 2>&1; echo "threads rc=$?"; grep -E "passed|failed" $O/threads.txt | tail -2
