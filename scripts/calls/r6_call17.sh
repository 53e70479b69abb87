#!/bin/bash
# round 6, GPU call 17: one host thread per shard for small batches: group tests, configs[2]-as-worded test, the 8-shard bench three ways
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6c17
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_configs.py tests/test_gpu_adapter.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -5 $O/pytest.txt
B="python bench.py --gpus 8 --same-device --workload criteo --features 33000000 --nnz 39 --rows 262144 --steps 2 --warmup 1"
FMX_GROUP_IN_STREAM=0 timeout 600 $B > $O/criteo_8shard_general.json 2> $O/e1.err; echo "rc=$?"
FMX_GROUP_THREADS=0 timeout 600 $B > $O/criteo_8shard_one_thread.json 2> $O/e2.err; echo "rc=$?"
timeout 600 $B > $O/criteo_8shard.json 2> $O/e3.err; echo "rc=$?"
python -c "
import json
for f in ('criteo_8shard_general', 'criteo_8shard_one_thread', 'criteo_8shard'):
    a=json.load(open('$O/%s.json' % f))
    print(f, a['value'], a['ms_per_step'], a['config']['batch'], a['phases_ms_per_batch'])"
nproc
