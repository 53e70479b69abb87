#!/bin/bash
# kernel-level profile of the per-rank compute side of the P=8 sharded step (scripts/gpu_shard_probe.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/shard8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $GRAFT_REPO_ROOT/scripts/gpu_shard_probe.py ${1:-1024} 8 > $OUT/probe.log 2>&1
grep "^chunk" $OUT/probe.log
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/s_kernel_stats.csv", recursive=True)
rows = list(csv.reader(open(f[0])))
for r in rows[1:10]:
    print("%-72s calls %5s avg %8.1f us total %8.1f ms" % (r[0][:72], r[1], float(r[3]) / 1e3, float(r[2]) / 1e6))
PY
