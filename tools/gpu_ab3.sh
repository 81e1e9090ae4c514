#!/bin/bash
# three-way A/B in one call: base, noslp, in-tree     usage: tools/gpu_ab3.sh <tag> libA libB
TAG=${1:-ab}; A=$2; B=$3
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2 3; do
  for v in $A $B new; do
    if [ $v = new ]; then unset ORBHIP_LIBRARY; else export ORBHIP_LIBRARY=$(pwd)/ab/liborbhip_$v.so; fi
    timeout 300 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io >> $OUT/bench_$v.jsonl 2>> $OUT/bench.err
  done
done
unset ORBHIP_LIBRARY
python3 - <<PY
import json
for f in ("$A","$B","new"):
    rows=[json.loads(l) for l in open("$OUT/bench_%s.jsonl"%f).read().strip().splitlines()]
    ks=rows[0]["kernels_ms_per_launch"].keys()
    print(f, [r["value"] for r in rows], "parity", [r["parity"]["mismatches"] for r in rows])
    print("   ", {k: round(sum(r["kernels_ms_per_launch"][k] for r in rows)/len(rows),4) for k in ks if rows[0]["kernels_ms_per_launch"][k]})
PY
tail -2 $OUT/bench.err
