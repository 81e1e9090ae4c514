#!/bin/bash
# round 5, fourth call: full GPU suite, config 5 on the FP4 matrix path (forms timed + full-DB parity), front-end loops (stereo, mono, RGB-D) and a traced stereo loop
TAG=${1:-r05_d}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
for f in i8 fp4x2 fp4x3 fp4x4 i8 fp4x3; do ORBHIP_NN=$f timeout 300 python tools/db_query_rate.py 2>> $OUT/nn.err | sed "s/^{/{\"ORBHIP_NN\": \"$f\", /" >> $OUT/nn_forms.jsonl; done
for f in fp4x3 fp4x4; do ORBHIP_NN=$f timeout 600 python tools/secondary_units.py --only config5 2>> $OUT/nn.err | sed "s/^{/{\"ORBHIP_NN\": \"$f\", /" >> $OUT/config5_fp4_parity.jsonl; done
timeout 600 python tools/dropin_loop_rate.py kitti euroc mono rgbd > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_loop_$TAG -o loop -- python $REPO/tools/dropin_loop_rate.py kitti > /dev/null 2>> $OUT/loop.err )
for f in $(find /tmp/prof_loop_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/loop_kernel_stats.csv; done
for f in $(find /tmp/prof_loop_$TAG -name "*kernel_trace.csv" | head -1); do cp $f $OUT/loop_kernel_trace.csv; done
for f in $(find /tmp/prof_loop_$TAG -name "*memory_copy_trace.csv" | head -1); do cp $f $OUT/loop_memory_copy_trace.csv; done
grep -E "concurrency|local_mapping|sequence\]|passed|failed|error|exit" $OUT/pytest_gpu.log | tail -14; cat $OUT/nn_forms.jsonl | cut -c1-330; cat $OUT/config5_fp4_parity.jsonl | cut -c1-900; cut -c1-1200 $OUT/dropin_loop.jsonl; tail -3 $OUT/nn.err $OUT/loop.err
