#!/bin/bash
# ORBHIP_SCHED x --streams at B = 512.  usage: tools/gpu_streams2.sh <tag>
TAG=${1:-st2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd "$(dirname "$0")/.."
for v in 0 1; do for st in 1 2 4; do
  ORBHIP_SCHED=$v timeout 300 python bench.py --steps 50 --warmup 3 --repeats 3 --batch 512 --streams $st --no-cpu-baseline --no-host-io > $OUT/bench_s${v}_st$st.json 2>> $OUT/bench.err
done; done
ORBHIP_SCHED=1 timeout 300 python bench.py --steps 30 --warmup 3 --repeats 3 --batch 1024 --streams 2 --no-cpu-baseline --no-host-io > $OUT/bench_s1_b1024_st2.json 2>> $OUT/bench.err
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_s*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v})
    except Exception as e: print(f, "failed", e)
PY
tail -3 $OUT/bench.err
