#!/bin/bash
# round 6, GPU call 7: first run of the XCD-resident epoch (fmx_xcd_kernels.h): its tests, the Criteo-shaped bench with and without it, hop trace
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c7
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_xcd.py -x -q -m gpu > $O/pytest_xcd.txt 2>&1; echo "xcd tests rc=$?"; tail -30 $O/pytest_xcd.txt
B="python bench.py --workload criteo --features 33000000 --nnz 39 --rows 1048576 --no-extras --no-cpu-baseline --steps 3 --warmup 1"
FMX_XCD=0 timeout 300 $B > $O/criteo_two_launches.json 2> $O/criteo_two_launches.err; echo "two launches rc=$?"
rm -f $O/hops.txt
FMX_XCD_TRACE=$O/hops.txt timeout 300 $B > $O/criteo_xcd.json 2> $O/criteo_xcd.err; echo "xcd rc=$?"
python - <<'PY'
import json
for f in ("criteo_two_launches", "criteo_xcd"):
    try:
        o = json.load(open("gpurun_out/r6c7/%s.json" % f))
        print(f, o["value"], o["ms_per_step"], o.get("batch"), o["roofline"].get("frac"))
    except Exception as ex:
        print(f, "failed", ex)
PY
head -5 $O/hops.txt; sed -n 40,60p $O/hops.txt
tail -3 $O/criteo_xcd.err
