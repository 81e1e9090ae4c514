#!/bin/bash
# schedule experiments: ORBHIP_SCHED variants at B = 512 (and 256), parity under the variant.  usage: tools/gpu_sched.sh <tag> "<sched values>"
TAG=${1:-sch}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd "$(dirname "$0")/.."
for v in ${2:-0 1}; do
  ORBHIP_SCHED=$v timeout 300 python -m pytest tests/test_full_size_gpu.py tests/test_parity_extract.py -m gpu -q -x 2>&1 | tail -1
  for b in 256 512; do ORBHIP_SCHED=$v timeout 300 python bench.py --steps 50 --warmup 3 --repeats 3 --batch $b --no-cpu-baseline --no-host-io > $OUT/bench_s${v}_b$b.json 2>> $OUT/bench.err; done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_s*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v})
    except Exception as e: print(f, "failed", e)
PY
tail -3 $OUT/bench.err
