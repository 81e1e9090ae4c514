#!/bin/bash
# round 5: GPU tests, the driver's bench command (now with matcher_calls / config5 / config4), the front-end loop with the stereo pair as one call
TAG=${1:-r05_b}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
timeout 300 python tools/dropin_loop_rate.py > $OUT/dropin_loop.jsonl 2>> $OUT/bench.err
timeout 200 python tools/matcher_latency.py > $OUT/matcher_latency.json 2>> $OUT/bench.err
timeout 100 python tools/single_frame_calls.py > $OUT/single_frame.txt 2>&1
grep -E "concurrency|dropin_loop|passed|failed|error|exit" $OUT/pytest_gpu.log | tail -12; cut -c1-400 $OUT/bench.json; tail -5 $OUT/bench.err; cut -c1-900 $OUT/dropin_loop.jsonl; cat $OUT/single_frame.txt | tail -5
