#!/bin/bash
# round 2, call D: all GPU tests, smoke, full-size config-5 parity, the default bench (host_io + cpu_baseline), rocprofv3 kernel stats
TAG=${1:-r2d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python tools/db_full_parity.py > $OUT/db_full_parity.json 2> $OUT/db_full_parity.err; echo "db parity exit $?" >> $OUT/db_full_parity.err
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
REPO=$(pwd)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python $REPO/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io > $REPO/$OUT/rocprof_bench.json 2> $REPO/$OUT/rocprof.err )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; done
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cat $OUT/db_full_parity.json; tail -2 $OUT/db_full_parity.err; cut -c1-2500 $OUT/bench.json; tail -2 $OUT/bench.err; head -12 $OUT/kernel_stats.csv
