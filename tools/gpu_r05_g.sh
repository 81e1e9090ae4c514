#!/bin/bash
# round 5, seventh call: the touched tests, matcher calls (FuseBatch with batched re-search, resident ComputeBoW), loops with and without staging helpers, traced mono / RGB-D loops
TAG=${1:-r05_g}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_projection.py tests/test_sequences.py tests/test_batch_matchers.py tests/test_reference_dropin.py -m gpu -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 600 python tools/secondary_units.py --only matcher_calls > $OUT/matcher_calls.json 2> $OUT/mc.err
timeout 600 python tools/dropin_loop_rate.py kitti mono rgbd > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
ORBHIP_COPY_THREADS=1 timeout 300 python tools/dropin_loop_rate.py kitti 2>> $OUT/loop.err | sed 's/^{/{"ORBHIP_COPY_THREADS": 1, /' >> $OUT/dropin_loop_ab.jsonl
timeout 300 python tools/dropin_loop_rate.py kitti 2>> $OUT/loop.err | sed 's/^{/{"ORBHIP_COPY_THREADS": "default", /' >> $OUT/dropin_loop_ab.jsonl
for w in mono rgbd; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_${w}_$TAG -o loop -- python $REPO/tools/dropin_loop_rate.py $w > /dev/null 2>> $OUT/loop.err )
  for f in $(find /tmp/prof_${w}_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/${w}_kernel_stats.csv; done
  for f in $(find /tmp/prof_${w}_$TAG -name "*kernel_trace.csv" | head -1); do cp $f $OUT/${w}_kernel_trace.csv; done
  for f in $(find /tmp/prof_${w}_$TAG -name "*memory_copy_trace.csv" | head -1); do cp $f $OUT/${w}_memory_copy_trace.csv; done
done
grep -E "passed|failed|error|exit" $OUT/pytest_gpu.log | tail -5; cut -c1-1500 $OUT/dropin_loop.jsonl; cut -c1-700 $OUT/dropin_loop_ab.jsonl; tail -3 $OUT/mc.err $OUT/loop.err
