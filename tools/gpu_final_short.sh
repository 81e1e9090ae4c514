#!/bin/bash
# The short form of tools/gpu_final.sh for a call of ~5 GPU-minutes: every GPU test, smoke, the default bench, rocprofv3 kernel stats of the bench
# command, the PMC passes, two larger batches, the two-rank run and a GPU fuzz - what changes when a kernel of the extraction chain changes.
# usage: tools/gpu_final_short.sh <tag>        (copy with tools/collect_profiles_short.sh <tag> <prefix>)
TAG=${1:-fins}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
REPO=$(pwd)
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
timeout 600 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python3 $REPO/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io > $REPO/$OUT/rocprof_bench.json 2> $REPO/$OUT/rocprof.err )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; done
bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1
for b in 256 1024 2048; do timeout 200 python3 bench.py --steps 30 --warmup 3 --repeats 3 --batch $b --no-cpu-baseline --no-host-io >> $OUT/bench_sweep.jsonl 2>> $OUT/bench.err; done
ORB_BENCH_SHARE_GPU=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 50 --warmup 3 --repeats 3 --batch 256 --no-cpu-baseline --no-host-io > $OUT/bench_2rank_shared.json 2>> $OUT/bench.err
timeout 300 python3 tests/test_fuzz_gpu.py 60 91 > $OUT/gpu_fuzz_60cases.txt 2>&1
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; cut -c1-1200 $OUT/bench.json; head -10 $OUT/kernel_stats.csv
python3 - <<PY
import json
for l in open('$OUT/bench_sweep.jsonl'):
    d=json.loads(l); print(d['config']['frames_per_step_per_gpu'], d['value'], d['ms_per_step'], d['parity']['mismatches'])
PY
tail -3 $OUT/bench.err; cut -c1-300 $OUT/bench_2rank_shared.json; tail -2 $OUT/gpu_fuzz_60cases.txt; ls $OUT/pmc | head -12
