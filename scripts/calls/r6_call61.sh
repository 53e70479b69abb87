#!/bin/bash
# round 6, GPU call 61: soak on the FINAL library (one-launch forms at k <= 32 too): the exchange tests 12 times, the fuzz file 8 times, in fresh processes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c61
mkdir -p $O
fail=0
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_gpu_small_one.py tests/test_gpu_parity.py tests/test_gpu_safety.py -q -m gpu -x -k "one_launch or conflict_free or long_runs or never_sees or gives_up or weight_side" > $O/soak_$i.txt 2>&1 || { fail=1; echo "soak $i FAILED"; tail -30 $O/soak_$i.txt; break; }
  rm -f $O/soak_$i.txt
done
echo "soak of the exchange tests: 12 rounds, fail=$fail"
for i in $(seq 1 8); do
  timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x > $O/fuzz_$i.txt 2>&1 || { echo "fuzz $i FAILED"; tail -30 $O/fuzz_$i.txt; break; }
  tail -1 $O/fuzz_$i.txt; rm -f $O/fuzz_$i.txt
done
