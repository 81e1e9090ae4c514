#!/bin/bash
# round 6, call T: the hand-ordered superstep with bounds shared between workgroups
TAG=${1:-r06_t}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_match.py -m gpu -q -x -k "brute or expanded or matrix_core" 2>&1 | tail -3 | tee $OUT/pytest.txt
for r in 1 2; do
for e in 1 0; do for sh in 1 0; do ORBHIP_NN_SHARE=$sh DB_EXPANDED=$e timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-140 | sed "s/^/expanded $e share $sh: /" | tee -a $OUT/rate.txt; done; done
ORBHIP_NN_BLOCK_VAR=64 DB_EXPANDED=1 timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-140 | sed "s/^/expanded 1 no kept pairs: /" | tee -a $OUT/rate.txt
done
exit 0
