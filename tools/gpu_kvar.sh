#!/bin/bash
# per-kernel standalone times (ORBHIP_SERIAL=1: one stream, nothing overlaps) of library variants   usage: tools/gpu_kvar.sh <tag> <variant>...
TAG=${1:-kvar}; shift
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for r in 1 2; do
  for v in "$@"; do
    if [ $v = new ]; then unset ORBHIP_LIBRARY; else export ORBHIP_LIBRARY=$REPO/ab/liborbhip_$v.so; fi
    ORBHIP_SERIAL=1 timeout 300 python3 bench.py --steps 30 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io --parity-slots 0 >> $OUT/bench_$v.jsonl 2>> $OUT/err.txt
  done
done
unset ORBHIP_LIBRARY
python3 - "$@" <<PY
import json,sys
for v in sys.argv[1:]:
    rows=[json.loads(l) for l in open("$OUT/bench_%s.jsonl"%v).read().strip().splitlines()]
    ks=[k for k in rows[0]["kernels_ms_per_launch"] if rows[0]["kernels_ms_per_launch"][k]]
    print(v, [r["value"] for r in rows], {k: round(sum(r["kernels_ms_per_launch"][k] for r in rows)/len(rows),4) for k in ks})
PY
tail -2 $OUT/err.txt
