#!/bin/bash
# A/B in one call: ORBHIP_LIBRARY=ab/liborbhip_<name>.so against the in-tree build, interleaved.   usage: tools/gpu_ab.sh <tag> <name> [bench args]
TAG=${1:-ab}; NAME=${2:-base}; shift; shift
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2 3; do
  ORBHIP_LIBRARY=$(pwd)/ab/liborbhip_$NAME.so timeout 300 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io "$@" >> $OUT/bench_A_$NAME.jsonl 2>> $OUT/bench.err
  timeout 300 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io "$@" >> $OUT/bench_B_new.jsonl 2>> $OUT/bench.err
done
python3 - <<PY
import json
for f in ("bench_A_$NAME","bench_B_new"):
    rows=[json.loads(l) for l in open("$OUT/%s.jsonl"%f).read().strip().splitlines()]
    ks=rows[0]["kernels_ms_per_launch"].keys()
    print(f, [r["value"] for r in rows], "parity", [r["parity"]["mismatches"] for r in rows])
    print("   ", {k: round(sum(r["kernels_ms_per_launch"][k] for r in rows)/len(rows),4) for k in ks if rows[0]["kernels_ms_per_launch"][k]})
PY
tail -2 $OUT/bench.err
