#!/bin/bash
# round 6, GPU call 46: k_small_one, the examples' row loads / stores non-temporal (default, 3) vs plain (0) vs plain loads + non-temporal stores (2)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" _nt0 _nt2; do
echo "libfmx$v.so"; FMX_LIB=$PWD/libfm_amd/libfmx$v.so timeout 300 python scripts/small_one_rate.py 2>&1 | grep "one launch"
done; done
