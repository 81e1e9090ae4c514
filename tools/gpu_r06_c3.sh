#!/bin/bash
# round 6: where does the FP4 scan's time go?  ablations (wrong results, timing only)
TAG=${1:-r06_c3}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for a in 0 3 7; do
  ORBHIP_NN_ABLATE=$a timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-260 | sed "s/^/ablate $a: /" | tee -a $OUT/ablate.txt
done
for f in; do ORBHIP_NN=$f timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-200 | sed "s/^/$f: /" | tee -a $OUT/ablate.txt; done
exit 0
