#!/bin/bash
# round 2, call A: GPU tests, smoke, host path sweep, default bench.  Everything lands in gpurun_out/$TAG.
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 400 python tools/host_io_rate.py 1.0 > $OUT/host_io.jsonl 2> $OUT/host_io.err
ORBHIP_COPY_THREADS=16 timeout 200 python - > $OUT/host_io_threads16.jsonl 2>> $OUT/host_io.err <<'PY'
import runpy, sys
sys.argv = ["host_io_rate.py", "0.5"]
runpy.run_path("tools/host_io_rate.py", run_name="__main__")
PY
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -5 $OUT/host_io.jsonl; tail -1 $OUT/host_io_threads16.jsonl; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err $OUT/host_io.err
