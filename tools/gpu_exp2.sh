#!/bin/bash
TAG=${1:-exp2}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_00_device.py tests/test_kernel_variants.py tests/test_parity_extract.py tests/test_full_size_gpu.py -x -q -m gpu -p no:cacheprovider -k "not config5" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
for NT in 1 2 4; do
  ORBHIP_BLUR_NT=$NT timeout 300 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io > $OUT/bench_nt$NT.json 2>> $OUT/bench.err
  ORBHIP_BLUR_NT=$NT timeout 300 python3 bench.py --batch 128 --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io > $OUT/bench_b128_nt$NT.json 2>> $OUT/bench.err
done
tail -4 $OUT/pytest.log
python3 - <<PY
import json
for f in ("bench_nt1","bench_nt2","bench_nt4","bench_b128_nt1","bench_b128_nt2","bench_b128_nt4"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d.get("parity",{}).get("mismatches"), {k:v for k,v in d["kernels_ms_per_launch"].items() if v})
    except Exception as e: print(f,"failed",e)
PY
tail -3 $OUT/bench.err
