#!/bin/bash
# ORBHIP_SCHED=3 (pyramid of call t+1 beside call t) against the default.  usage: tools/gpu_sched3.sh <tag>
TAG=${1:-s4}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_parity_match.py -m gpu -q -x -k "device_pipeline" 2>&1 | tail -1
for v in 0 3; do for b in 128 256 512; do
  ORBHIP_SCHED=$v timeout 300 python bench.py --steps 60 --warmup 4 --repeats 3 --batch $b --no-cpu-baseline --no-host-io > $OUT/bench_s${v}_b$b.json 2>> $OUT/bench.err
done; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_s*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["check"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v})
    except Exception as e: print(f, "failed", e)
PY
tail -3 $OUT/bench.err
