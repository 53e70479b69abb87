#!/bin/bash
# round 5, GPU call 15: the multi-GPU code path as typed, eight shards on the one GPU (loopback exchange), both exchange forms
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5c15
mkdir -p $O
timeout 400 python bench.py --gpus 8 --same-device --steps 3 --warmup 1 > $O/bench_8_shards_one_device.json 2> $O/bench8.err
timeout 400 python bench.py --gpus 8 --same-device --steps 3 --warmup 1 --exchange rsag > $O/bench_8_shards_one_device_rsag.json 2>> $O/bench8.err
timeout 400 python bench.py --gpus 4 --same-device --steps 3 --warmup 1 --pipeline > $O/bench_4_shards_one_device_pipeline.json 2>> $O/bench8.err
for f in bench_8_shards_one_device bench_8_shards_one_device_rsag bench_4_shards_one_device_pipeline; do python -c "
import json; o=json.load(open('$O/$f.json')); print('$f', o['value'], o['n_gpus'], o['exchange'].get('algo'), o.get('phases_ms_per_batch'))"; done
