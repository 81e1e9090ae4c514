#!/bin/bash
# VALU blur vs matrix-core blur: parity and standalone / production timings.  usage: tools/gpu_blur.sh <tag>
TAG=${1:-bl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd "$(dirname "$0")/.."
for v in valu mfma; do
  ORBHIP_BLUR=$v timeout 300 python -m pytest tests/test_full_size_gpu.py tests/test_parity_extract.py tests/test_reference_extractor.py tests/test_golden_frames.py -m gpu -q -x 2>&1 | tail -1
  ORBHIP_BLUR=$v ORBHIP_SERIAL=1 timeout 300 python bench.py --steps 20 --warmup 3 --repeats 3 --batch 256 --no-cpu-baseline --no-host-io > $OUT/bench_${v}_serial_b256.json 2>> $OUT/bench.err
  ORBHIP_BLUR=$v timeout 300 python bench.py --steps 50 --warmup 3 --repeats 3 --batch 512 --no-cpu-baseline --no-host-io > $OUT/bench_${v}_b512.json 2>> $OUT/bench.err
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v})
    except Exception as e: print(f, "failed", e)
PY
tail -3 $OUT/bench.err
