#!/bin/bash
# One gpurun call of round 4: GPU tests -> smoke -> bench (the driver's command) -> rocprofv3 kernel stats of the same command -> front-end loop
# rate + its kernel trace -> single-frame call timeline.  Everything lands in gpurun_out/$TAG; tools/gpu_pmc.sh collects the counter passes.
TAG=${1:-r04}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; numactl -H 2>/dev/null | head -4; for d in /sys/class/drm/card*/device/numa_node; do echo $d $(cat $d); done ) > $OUT/box.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop > $OUT/rocprof_bench.json 2> $OUT/rocprof.err )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/bench_kernel_stats.csv; done
timeout 300 python tools/dropin_loop_rate.py > $OUT/dropin_loop.jsonl 2>> $OUT/bench.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_loop_$TAG -o loop -- python $REPO/tools/dropin_loop_rate.py kitti > /dev/null 2>> $OUT/rocprof.err )
for f in $(find /tmp/prof_loop_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/loop_kernel_stats.csv; done
for f in $(find /tmp/prof_loop_$TAG -name "*kernel_trace.csv" | head -1); do cp $f $OUT/loop_kernel_trace.csv; done
for f in $(find /tmp/prof_loop_$TAG -name "*memory_copy_trace.csv" | head -1); do cp $f $OUT/loop_memory_copy_trace.csv; done
timeout 100 python tools/single_frame_calls.py > $OUT/single_frame.txt 2>&1
timeout 200 python tools/matcher_latency.py > $OUT/matcher_latency.json 2>> $OUT/bench.err
ORBHIP_SERIAL=1 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop > $OUT/bench_serial_b512.json 2>> $OUT/bench.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_single_$TAG -o s -- python $REPO/tools/single_frame_calls.py > /dev/null 2>> $OUT/rocprof.err )
for f in $(find /tmp/prof_single_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/single_frame_kernel_stats.csv; done
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cut -c1-600 $OUT/bench.json; tail -6 $OUT/bench.err; cat $OUT/single_frame.txt; cat $OUT/matcher_latency.json; cat $OUT/bench_kernel_stats.csv | cut -c1-140 | head -14
