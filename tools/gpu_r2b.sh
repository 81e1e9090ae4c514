#!/bin/bash
TAG=${1:-r2b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python tools/host_io_rate.py 0.8 > $OUT/host_io.jsonl 2> $OUT/host_io.err
timeout 200 tools/ubench > $OUT/ubench.txt 2>&1
tail -6 $OUT/pytest_gpu.log; cat $OUT/host_io.jsonl; tail -3 $OUT/host_io.err; tail -60 $OUT/ubench.txt
