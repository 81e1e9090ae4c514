#!/bin/bash
# round 5, fifth call: FP4 scan shapes (tile skip, rows per workgroup), NN parity tests, sequences with the ORBvoc-shaped vocabulary, ComputeBoW after the sorting network
TAG=${1:-r05_e}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_match.py tests/test_bow.py tests/test_sequences.py -m gpu -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
for f in i8 fp4:4:2:13 fp4:4:2:15 fp4:4:2:16 fp4:2:3:15 fp4:2:3:16 fp4:3:2:15 fp4:2:2:15 fp4:4:2:15; do ORBHIP_NN=$f timeout 300 python tools/db_query_rate.py 2>> $OUT/nn.err | sed "s/^{/{\"ORBHIP_NN\": \"$f\", /" >> $OUT/nn_forms.jsonl; done
ORBHIP_NN=fp4:4:2:15 timeout 600 python tools/secondary_units.py --only config5 2>> $OUT/nn.err | sed "s/^{/{\"ORBHIP_NN\": \"fp4:4:2:15\", /" >> $OUT/config5_fp4_parity.jsonl
timeout 600 python tools/dropin_loop_rate.py mono rgbd > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
timeout 200 python tools/bow_rate.py > $OUT/bow_rate.json 2>> $OUT/loop.err
timeout 600 python tools/secondary_units.py --only matcher_calls > $OUT/matcher_calls.json 2>> $OUT/loop.err
grep -E "passed|failed|error|exit" $OUT/pytest_gpu.log | tail -5; cat $OUT/nn_forms.jsonl | cut -c1-220; cat $OUT/config5_fp4_parity.jsonl | cut -c200-1000; cut -c1-1500 $OUT/dropin_loop.jsonl; cat $OUT/bow_rate.json; tail -n 3 $OUT/nn.err $OUT/loop.err
