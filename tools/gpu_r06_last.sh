#!/bin/bash
# round 6, the last check of the tree as committed: every GPU test, smoke, the driver's bench command
TAG=${1:-r06_last}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
SECONDS=0
timeout 1200 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? after $SECONDS s" >> $OUT/bench.err
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; cut -c1-400 $OUT/bench.json; tail -1 $OUT/bench.err
exit 0
