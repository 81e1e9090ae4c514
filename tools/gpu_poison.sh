#!/bin/bash
# pytest -m gpu with every device allocation of the library pre-filled with a byte pattern (ORBHIP_POISON): a kernel that reads memory it
# never wrote (and so depends on what a previous tenant of the box left there) fails on any box.   usage: tools/gpu_poison.sh <tag>
TAG=${1:-poison}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG
mkdir -p $OUT
for P in 255 127 1; do
  ORBHIP_POISON=$P timeout 900 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/pytest_poison$P.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_poison$P.log
  ORBHIP_POISON=$P timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > $OUT/smoke_poison$P.log 2>&1; echo "smoke exit $?" >> $OUT/smoke_poison$P.log
  tail -4 $OUT/pytest_poison$P.log; tail -2 $OUT/smoke_poison$P.log
done
