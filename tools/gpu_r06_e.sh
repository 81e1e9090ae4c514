#!/bin/bash
# round 6, call E: the host path's rate inside a process that has held (and run, and destroyed) a resident context - by stream-priority mode and creation order
TAG=${1:-r06_e}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
python tools/host_io_order_experiment.py plain >> $OUT/host_io_order.jsonl 2>> $OUT/err.txt
for p in 2 1 0; do
  ORBHIP_STREAM_PRIO=$p python tools/host_io_order_experiment.py big-ran-closed >> $OUT/host_io_order.jsonl 2>> $OUT/err.txt
done
ORBHIP_STREAM_PRIO=2 python tools/host_io_order_experiment.py hostfirst-big-ran-closed >> $OUT/host_io_order.jsonl 2>> $OUT/err.txt
ORBHIP_STREAM_PRIO=2 python tools/host_io_order_experiment.py big-closed >> $OUT/host_io_order.jsonl 2>> $OUT/err.txt
done
cat $OUT/host_io_order.jsonl; tail -3 $OUT/err.txt
exit 0
