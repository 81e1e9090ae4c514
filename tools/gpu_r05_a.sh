#!/bin/bash
# round 5, first call: GPU tests (new: three-thread re-entrancy, large-frame grid), smoke, matcher call latencies after the arena / per-thread stream change
TAG=${1:-r05_a}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 200 python tools/matcher_latency.py > $OUT/matcher_latency.json 2> $OUT/matcher_latency.err
grep -E "concurrency|passed|failed|error|exit" $OUT/pytest_gpu.log | tail -12; tail -2 $OUT/smoke.log; cat $OUT/matcher_latency.json; tail -3 $OUT/matcher_latency.err
