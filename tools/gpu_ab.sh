#!/bin/bash
# A/B of one environment switch at B = 256 and 512.  usage: tools/gpu_ab.sh <tag> <VAR> "<values>"
TAG=${1:-ab}; VAR=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd "$(dirname "$0")/.."
for v in $3; do for b in 256 512; do
  env $VAR=$v timeout 300 python bench.py --steps 60 --warmup 4 --repeats 3 --batch $b --no-cpu-baseline --no-host-io > $OUT/bench_${v}_b$b.json 2>> $OUT/bench.err
done; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v})
    except Exception as e: print(f, "failed", e)
PY
tail -2 $OUT/bench.err
