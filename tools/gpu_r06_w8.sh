#!/bin/bash
# round 6: config 5's main pass with eight wavefronts per workgroup (ORBHIP_NN_WAVES=8: one workgroup per CU, 1024 queries per staged tile) against four
TAG=${1:-r06_w8}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ORBHIP_NN_WAVES=8 timeout 900 python -m pytest tests/test_parity_match.py -m gpu -q -x -k "brute or expanded or matrix_core" 2>&1 | tail -3 | tee $OUT/pytest.txt
ORBHIP_NN_WAVES=8 timeout 600 python tools/nn_size_fuzz.py 30 99 2>&1 | tail -2 | tee $OUT/fuzz.txt
for r in 1 2 3; do
for e in 1 0; do for w in 8 4; do ORBHIP_NN_WAVES=$w DB_EXPANDED=$e timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-140 | sed "s/^/expanded $e waves $w: /" | tee -a $OUT/rate.txt; done; done
done
for w in 8 4; do ORBHIP_NN_STATS=1 ORBHIP_NN_WAVES=$w DB_EXPANDED=1 timeout 120 python tools/db_query_rate.py 2>&1 | grep "kept" | tail -1 | sed "s/^/waves $w: /" | tee -a $OUT/rate.txt; done
for n in 100 1000 3000; do for w in 8 4; do ORBHIP_NN_WAVES=$w DB_EXPANDED=1 timeout 300 python tools/db_query_rate.py $n 2>&1 | tail -1 | cut -c1-140 | sed "s/^/keyframes $n waves $w: /" | tee -a $OUT/rate.txt; done; done
exit 0
