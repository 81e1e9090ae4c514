#!/bin/bash
TAG=${1:-nn}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_full_size_gpu.py tests/test_parity_match.py tests/test_sharding.py tests/test_host_pipeline.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
for r in 1 2; do
  ORBHIP_LIBRARY=$REPO/ab/liborbhip_base.so timeout 300 python3 tools/db_query_rate.py >> $OUT/db_query.jsonl 2>> $OUT/err.txt
  timeout 300 python3 tools/db_query_rate.py >> $OUT/db_query.jsonl 2>> $OUT/err.txt
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nnprof -o nn -- python3 $REPO/tools/db_query_rate.py > /dev/null 2>> $OUT/err.txt ); f=$(find /tmp/nnprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/nn_kernel_stats.csv
tail -3 $OUT/pytest.log; cat $OUT/db_query.jsonl; head -5 $OUT/nn_kernel_stats.csv; tail -3 $OUT/err.txt
