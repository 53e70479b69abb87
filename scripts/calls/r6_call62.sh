#!/bin/bash
# round 6, GPU call 62: runs forced on rows that conflict everywhere (FMX_SEQ_RUNS=1: runs of one to a few rows, empty rows, repeated ids, rows beyond the register path)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "runs_forced" 2>&1 | tail -15
