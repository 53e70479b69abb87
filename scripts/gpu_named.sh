#!/bin/bash
# named configs of BASELINE.json on one GPU, one-pass batch rule, batch-size sweep
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('   %.1f M ex/s  frac %.3f  deferred/ex %.2f  launch %.3f ms' % (d['value']/1e6, r['frac'], r.get('deferred_features_per_example',-1), r['avg_launch_ms']))
"; }
for b in 16384 32768 65536 131072 262144; do
  echo "C2 n=1e7 k=32 nnz=16 rows=8M batch=$b"; timeout 200 $B --features 10000000 --factors 32 --nnz 16 --rows 8388608 --batch $b 2>/dev/null | show
done
for b in 32768 65536 131072 262144; do
  echo "C3 n=3.3e7 k=64 nnz=39 rows=4M batch=$b"; timeout 200 $B --features 33000000 --factors 64 --nnz 39 --rows 4194304 --batch $b 2>/dev/null | show
done
for b in 65536 131072 262144; do
  echo "NS n=1e8 k=64 nnz=32 rows=4M batch=$b"; timeout 200 $B --batch $b 2>/dev/null | show
done
