#!/bin/bash
# round 5, sixth call: FP4 scan with several tiles per barrier; stereo constructor after the critical-path work; ComputeBoW after the pipelined norm loop
TAG=${1:-r05_f}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_match.py tests/test_bow.py tests/test_parity_stereo.py tests/test_dropin_loop.py tests/test_dropin_cpp.py -m gpu -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
for f in i8 fp4:4:2:15:1 fp4:4:2:15:2 fp4:4:2:15:4 fp4:4:2:15:8 fp4:2:3:15:4 fp4:2:2:15:4 fp4:4:2:13:4 fp4:4:2:16:4 fp4:3:2:15:4 fp4:4:2:15:4; do ORBHIP_NN=$f timeout 300 python tools/db_query_rate.py 2>> $OUT/nn.err | sed "s/^{/{\"ORBHIP_NN\": \"$f\", /" >> $OUT/nn_forms.jsonl; done
timeout 600 python tools/dropin_loop_rate.py kitti euroc > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
timeout 200 python tools/bow_rate.py > $OUT/bow_rate.json 2>> $OUT/loop.err
timeout 600 python tools/secondary_units.py --only matcher_calls > $OUT/matcher_calls.json 2>> $OUT/loop.err
grep -E "dropin_loop|passed|failed|error|exit" $OUT/pytest_gpu.log | tail -8 | cut -c1-400; cat $OUT/nn_forms.jsonl | cut -c1-160; cut -c1-700 $OUT/dropin_loop.jsonl; cat $OUT/bow_rate.json; tail -n 3 $OUT/nn.err $OUT/loop.err
