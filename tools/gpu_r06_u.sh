#!/bin/bash
# round 6, call U: kept-pair statistics of the FP4 scan with and without shared bounds
TAG=${1:-r06_u}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for sh in 1 0; do ORBHIP_NN_STATS=1 ORBHIP_NN_SHARE=$sh DB_EXPANDED=1 timeout 300 python tools/db_query_rate.py 2>&1 | tail -4 | cut -c1-140 | sed "s/^/share $sh: /" | tee -a $OUT/rate.txt; done
exit 0
