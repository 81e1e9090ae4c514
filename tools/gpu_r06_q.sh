#!/bin/bash
# round 6, call Q: config 5 with the hand-ordered superstep (k_hamming_nn_fp4b) against the compiler-scheduled tile loop
TAG=${1:-r06_q}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_match.py -m gpu -q -x -k "brute or expanded or matrix_core" 2>&1 | tail -3 | tee $OUT/pytest.txt
for b in 1 0 1 0; do for e in 0 1; do ORBHIP_NN_BLOCK=$b DB_EXPANDED=$e timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-420 | sed "s/^/block $b expanded $e: /" | tee -a $OUT/rate.txt; done; done
exit 0
