#!/bin/bash
# round 6, GPU call 45: smoke() with the runs leg; soak of the new in-launch exchanges (k_run_fused, k_small_one): their tests 25 times over, the fuzz file 6 times
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c45
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
fail=0
for i in $(seq 1 25); do
  timeout 600 python -m pytest tests/test_gpu_small_one.py tests/test_gpu_parity.py tests/test_gpu_safety.py -q -m gpu -x -k "one_launch or conflict_free or long_runs or never_sees or gives_up or weight_side" > $O/soak_$i.txt 2>&1 || { fail=1; echo "soak $i FAILED"; tail -30 $O/soak_$i.txt; break; }
  rm -f $O/soak_$i.txt
done
echo "soak of the exchange tests: 25 rounds, fail=$fail"
for i in $(seq 1 6); do
  timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x > $O/fuzz_$i.txt 2>&1 || { echo "fuzz $i FAILED"; tail -30 $O/fuzz_$i.txt; break; }
  tail -1 $O/fuzz_$i.txt; rm -f $O/fuzz_$i.txt
done
