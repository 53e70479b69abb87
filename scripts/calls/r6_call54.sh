#!/bin/bash
# round 6, GPU call 54: sanity of the final build (after the reverted experiment of call 53): smoke(), the exchange tests, both rate scripts
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_small_one.py tests/test_gpu_parity.py tests/test_gpu_safety.py -q -m gpu -x -k "one_launch or conflict_free or long_runs or never_sees or gives_up or weight_side" 2>&1 | tail -2
timeout 300 python scripts/small_one_rate.py 2>&1 | grep launch
timeout 300 python scripts/seq_rate.py 2>&1 | grep examples
