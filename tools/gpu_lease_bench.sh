#!/bin/bash
# One fresh-lease acceptance run (tools/gpu_repro.sh) followed by the default bench and rocprofv3 kernel stats of the bench command.   usage: tools/gpu_lease_bench.sh <tag>
TAG=${1:-lb}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=gpurun_out/$TAG
bash tools/gpu_repro.sh $TAG > /dev/null 2>&1
export TMPDIR=/tmp
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python3 $REPO/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io > $REPO/$OUT/rocprof_bench.json 2> $REPO/$OUT/rocprof.err )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; done
tail -3 $OUT/pytest.log; tail -2 $OUT/smoke.log | cut -c1-160; cut -c1-400 $OUT/bench.json; head -8 $OUT/kernel_stats.csv
