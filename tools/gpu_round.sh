#!/bin/bash
# One gpurun call: GPU tests -> smoke -> bench -> rocprofv3 kernel trace.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
for cfg in "64 1" "128 1" "256 1" "512 1" "1024 1" "256 2"; do set -- $cfg; timeout 200 python bench.py --steps 20 --warmup 3 --batch $1 --streams $2 --no-cpu-baseline >> $OUT/bench_sweep.jsonl 2>> $OUT/bench.err; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/rocprof_bench.json 2> $OLDPWD/$OUT/rocprof.err )
find /tmp/prof_$TAG -name "*stats*" -o -name "*kernel_trace*" | head -20 > $OUT/prof_files.txt
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -2); do cp $f $OUT/; done
for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1); do head -400 $f > $OUT/kernel_trace_head.csv; done
ORB_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 32 > $OUT/bench_2rank_shared.json 2>> $OUT/bench.err; cat $OUT/bench_2rank_shared.json | cut -c1-300
timeout 120 python tools/host_io_rate.py > $OUT/host_io.json 2>> $OUT/bench.err; cat $OUT/host_io.json
timeout 200 python tools/stereo_rate.py > $OUT/stereo_rate.json 2>> $OUT/bench.err; cat $OUT/stereo_rate.json
timeout 200 python tools/db_query_rate.py > $OUT/db_query.json 2>> $OUT/bench.err; cat $OUT/db_query.json
ORBHIP_SERIAL=1 timeout 200 python bench.py --steps 20 --warmup 3 --batch 256 --no-cpu-baseline > $OUT/bench_serial_b256.json 2>> $OUT/bench.err
timeout 200 python tools/matcher_latency.py > $OUT/matcher_latency.json 2>> $OUT/bench.err
timeout 300 python tools/bow_rate.py > $OUT/bow_rate.json 2>> $OUT/bench.err
timeout 300 python tools/camera_rate.py > $OUT/camera_rate.json 2>> $OUT/bench.err
timeout 120 tools/ubench > $OUT/ubench.txt 2>&1; cat $OUT/ubench.txt
tail -5 $OUT/pytest_gpu.log; tail -3 $OUT/smoke.log; cat $OUT/bench.json; python - <<PY
import json
for l in open('$OUT/bench_sweep.jsonl'):
    d=json.loads(l); print(d['config']['frames_per_step_per_gpu'], d['config']['streams_per_gpu'], d['value'], d['kernels_ms_per_launch'])
PY
 cat $OUT/*kernel_stats.csv | head -30; tail -3 $OUT/bench.err; ls $OUT
