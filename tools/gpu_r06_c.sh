#!/bin/bash
# round 6, call C: config 5 (the descriptor-database scan): form tests, the full 20 M x 2000 parity, the query rate seeded / unseeded.   usage: tools/gpu_r06_c.sh <tag>
TAG=${1:-r06_c}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_match.py tests/test_full_size_gpu.py -m gpu -q -x -k "brute_force or db or nn or config5" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
timeout 300 python tools/db_query_rate.py > $OUT/db_query_rate.jsonl 2>> $OUT/tools.err
ORBHIP_NN_SEED=0 timeout 300 python tools/db_query_rate.py > $OUT/db_query_rate_unseeded.jsonl 2>> $OUT/tools.err
tail -4 $OUT/pytest.log; cat $OUT/db_query_rate.jsonl | cut -c1-600; echo; cat $OUT/db_query_rate_unseeded.jsonl | cut -c1-600; tail -3 $OUT/tools.err
exit 0
