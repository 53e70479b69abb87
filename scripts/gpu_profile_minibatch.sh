#!/bin/bash
# rocprofv3 kernel stats of (a) the exact MINIBATCH epoch on one GPU and (b) the per-rank compute side of the P=8 sharded step
OUT=$GRAFT_REPO_ROOT/gpurun_out/mb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mb -o mb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mode minibatch --batch 131072 --w0-chunk 1024 > $OUT/mb.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shard8 -o s -- python $GRAFT_REPO_ROOT/scripts/gpu_shard_probe.py 1024 8 64 262144 > $OUT/shard8.log 2>&1
grep "chunk=" $OUT/shard8.log
tail -1 $OUT/mb.json | cut -c1-200
