#!/bin/bash
# experiment: w_j co-located behind its factor row in padded rows (FMX_ROW_PAD floats) vs the separate w[] array
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rowpad
for v in "" pad16 pad32 pad64 ""; do
  L=$GRAFT_REPO_ROOT/libfm_amd/libfmx${v:+_$v}.so
  echo "== ${v:-base}"
  FMX_LIB=$L timeout 300 python bench.py --no-cpu-baseline --steps 8 --rows 4194304 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  train %.1f M ex/s  ms/step %.3f  predict %.1f M rows/s  hogwild %.1f' % (d['value']/1e6, d['ms_per_step'], d['predict']['value']/1e6, d['hogwild']['value']/1e6))"
done 2>&1 | tee gpurun_out/rowpad/result.txt
