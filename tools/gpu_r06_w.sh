#!/bin/bash
# round 6, call W: the whole GPU suite + the secondary units (config 5 on the registered database) on the tree with the hand-ordered FP4 scan
TAG=${1:-r06_w}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python tools/db_full_parity.py > $OUT/db_full_parity.json 2> $OUT/db_full_parity.err
for e in 0 1; do DB_EXPANDED=$e timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 >> $OUT/db_query_rate.jsonl; done
tail -4 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; cut -c1-600 $OUT/db_full_parity.json; cut -c1-300 $OUT/db_query_rate.jsonl
exit 0
