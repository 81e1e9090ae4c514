#!/bin/bash
TAG=${1:-nn2}; shift
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for r in 1 2; do
  for v in "$@"; do
    if [ $v = new ]; then unset ORBHIP_LIBRARY; else export ORBHIP_LIBRARY=$REPO/ab/liborbhip_$v.so; fi
    timeout 300 python3 tools/db_query_rate.py >> $OUT/db_query.jsonl 2>> $OUT/err.txt
  done
done
python3 - <<PY
import json
for l in open("$OUT/db_query.jsonl"): d=json.loads(l); print(d["library"].split("/")[-1], d["query_ms"], d["mfma_i8_TOPs"])
PY
tail -2 $OUT/err.txt
