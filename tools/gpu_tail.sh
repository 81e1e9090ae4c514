#!/bin/bash
# fused pyramid tail on / off: parity and timings.  usage: tools/gpu_tail.sh <tag>
TAG=${1:-tl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_kernel_variants.py tests/test_full_size_gpu.py tests/test_parity_extract.py -m gpu -q -x 2>&1 | tail -1
for v in 1 0; do
  ORBHIP_PYR_TAIL=$v ORBHIP_SERIAL=1 timeout 300 python bench.py --steps 20 --warmup 3 --repeats 3 --batch 256 --no-cpu-baseline --no-host-io > $OUT/bench_tail${v}_serial_b256.json 2>> $OUT/bench.err
  ORBHIP_PYR_TAIL=$v timeout 300 python bench.py --steps 50 --warmup 3 --repeats 3 --batch 512 --no-cpu-baseline --no-host-io > $OUT/bench_tail${v}_b512.json 2>> $OUT/bench.err
  ORBHIP_PYR_TAIL=$v timeout 300 python bench.py --steps 50 --warmup 3 --repeats 3 --batch 128 --no-cpu-baseline --no-host-io > $OUT/bench_tail${v}_b128.json 2>> $OUT/bench.err
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v})
    except Exception as e: print(f, "failed", e)
PY
tail -3 $OUT/bench.err
