#!/bin/bash
# round 6, call D: the N > 1 path on one GPU (eight ranks + extras), then the driver's own command (N = 1, default flags)
TAG=${1:-r06_d}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sharding.py -m gpu -q -x -s > $OUT/pytest_sharding.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_sharding.log
SECONDS=0
timeout 1500 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? after $SECONDS s" >> $OUT/bench.err
tail -5 $OUT/pytest_sharding.log; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
exit 0
