#!/bin/bash
# round 6, call L: matcher members after the Fuse changes (membership read with the point, distance test first): parity suites + member timings
TAG=${1:-r06_l}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_reference_dropin.py tests/test_batch_matchers.py tests/test_projection_poses.py tests/test_dropin_loop.py tests/test_secondary_units.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 900 python bench.py --no-traffic --no-cpu-baseline --no-host-io --steps 10 --repeats 1 --parity-slots 4 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
for k, v in d["matcher_calls"]["members"].items(): print(f"{k[:66]:66s} {v['gpu_ms']:.3f} ({v['gpu_ms_inside_the_library']:.3f}) ref {v['ref_ms']:.3f} {v['parity']}")
for k, v in d["matcher_calls"]["back_end_loops"].items(): print(k, v)
print(d["dropin_loop"]["ms_per_frame_gpu"], d["dropin_loop"]["parity"])
PY
exit 0
