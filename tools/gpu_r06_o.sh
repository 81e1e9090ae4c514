#!/bin/bash
# round 6, call O: config 5 with the database expanded once in HBM (LDS-DMA staging)
TAG=${1:-r06_o}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_match.py -m gpu -q -x -k "expanded or matrix_core" 2>&1 | tail -2 | tee $OUT/pytest.txt
for e in 0 1 0 1; do DB_EXPANDED=$e timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-420 | sed "s/^/expanded $e: /" | tee -a $OUT/rate.txt; done
exit 0
