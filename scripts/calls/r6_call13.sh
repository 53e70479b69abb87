#!/bin/bash
# round 6, GPU call 13: BASELINE configs[2] as worded -- Criteo-shaped rows, V sharded over 8 feature shards (one GPU: the loopback exchange)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c13
mkdir -p $O
timeout 600 python bench.py --gpus 8 --same-device --workload criteo --features 33000000 --nnz 39 --rows 131072 --steps 2 --warmup 1 > $O/criteo_8shard.json 2> $O/criteo_8shard.err; echo "rc=$?"
python -c "
import json
a=json.load(open('$O/criteo_8shard.json'))
print(a['value'], a['ms_per_step'], a['config']['batch'], a['phases_ms_per_batch'])"
tail -3 $O/criteo_8shard.err
