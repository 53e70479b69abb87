#!/bin/bash
cd $GRAFT_REPO_ROOT
for spw in 8 32; do echo "== FMX_P2_SPW=$spw"; FMX_P2_SPW=$spw bash scripts/gpu_sgd_trace.sh | grep "k_apply_seg\|k_fused"; done
