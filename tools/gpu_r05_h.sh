#!/bin/bash
# round 5, eighth call: bag of words on the breadth-first tree records / one-pass sorting network / one-copy fetch: tests, rates, traced mono loop
TAG=${1:-r05_h}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bow.py tests/test_sequences.py tests/test_batch_matchers.py tests/test_reference_matchers.py tests/test_dropin_cpp.py -m gpu -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python tools/bow_rate.py > $OUT/bow_rate.json 2> $OUT/bow.err
timeout 600 python tools/dropin_loop_rate.py mono rgbd > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
timeout 600 python tools/secondary_units.py --only matcher_calls > $OUT/matcher_calls.json 2> $OUT/mc.err
for w in mono; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_${w}_$TAG -o loop -- python $REPO/tools/dropin_loop_rate.py $w > /dev/null 2>> $OUT/loop.err )
  for f in $(find /tmp/prof_${w}_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/${w}_kernel_stats.csv; done
  for f in $(find /tmp/prof_${w}_$TAG -name "*kernel_trace.csv" | head -1); do cp $f $OUT/${w}_kernel_trace.csv; done
  for f in $(find /tmp/prof_${w}_$TAG -name "*memory_copy_trace.csv" | head -1); do cp $f $OUT/${w}_memory_copy_trace.csv; done
done
grep -E "passed|failed|error|exit" $OUT/pytest_gpu.log | tail -5; cut -c1-900 $OUT/bow_rate.json; cut -c1-900 $OUT/dropin_loop.jsonl; head -12 $OUT/mono_kernel_stats.csv | cut -d, -f1-4
exit 0
