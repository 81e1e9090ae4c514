#!/bin/bash
# kernel / copy timeline of single-frame drop-in calls (rocprofv3 --kernel-trace --memory-copy-trace --stats)
TAG=${1:-single}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/single_frame_calls.py > $OUT/plain.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_$TAG -o s -- python3 $REPO/tools/single_frame_calls.py > $OUT/prof.txt 2>&1 )
for f in $(find /tmp/prof_$TAG -name "*_stats.csv"); do cp $f $OUT/; done
for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1); do tail -60 $f | cut -d, -f8-20 > $OUT/kernel_trace_tail.csv; done
cat $OUT/plain.txt $OUT/prof.txt | grep single_frame; cat $OUT/*kernel_stats.csv | cut -c1-160; cat $OUT/*memory_copy_stats.csv 2>/dev/null | cut -c1-160
