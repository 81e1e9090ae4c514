#!/bin/bash
# round 6, call X: rows per workgroup of config 5's main pass sized to whole rounds of the chip's workgroup slots (ORBHIP_NN_BALANCE=0: chunks of 2^15 rows)
TAG=${1:-r06_x}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_match.py -m gpu -q -x -k "brute or expanded or matrix_core" 2>&1 | tail -3 | tee $OUT/pytest.txt
for r in 1 2 3; do
for e in 1 0; do for b in 1 0; do ORBHIP_NN_BALANCE=$b DB_EXPANDED=$e timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-140 | sed "s/^/expanded $e balance $b: /" | tee -a $OUT/rate.txt; done; done
done
for n in 100 1000 3000; do for b in 1 0; do ORBHIP_NN_BALANCE=$b DB_EXPANDED=1 timeout 300 python tools/db_query_rate.py $n 2>&1 | tail -1 | cut -c1-140 | sed "s/^/keyframes $n balance $b: /" | tee -a $OUT/rate.txt; done; done
exit 0
