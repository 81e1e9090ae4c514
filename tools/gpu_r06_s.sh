#!/bin/bash
# round 6, call S: text / staging variants of the hand-ordered superstep (ORBHIP_NN_BLOCK_VAR: 1 no tests, 2 maxima only, 3 matrix instructions alone, +4 no staging and no barrier)
TAG=${1:-r06_s}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in ${VARS:-0 1 2 3 4 5 7 0 1 2 3 4 5 7}; do ORBHIP_NN_BLOCK_VAR=$v DB_EXPANDED=1 timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-140 | sed "s/^/var $v: /" | tee -a $OUT/rate.txt; done
exit 0
