#!/bin/bash
# round 6, call A: every GPU test, smoke, the secondary units (matcher_calls with the projection on the device), the front-end loops.   usage: tools/gpu_r06_a.sh <tag>
TAG=${1:-r06_a}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rs -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python tools/secondary_units.py --blur-round-mode 1 > $OUT/secondary.json 2> $OUT/secondary.err
timeout 600 python tools/dropin_loop_rate.py kitti euroc mono rgbd > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
tail -6 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; python - <<PY
import json
try:
    s = json.load(open("$OUT/secondary.json"))
    for k, v in s["matcher_calls"].items():
        print(k, json.dumps(v)[:260])
except Exception as e:
    print("secondary:", e); print(open("$OUT/secondary.err").read()[-1500:])
PY
cut -c1-700 $OUT/dropin_loop.jsonl; tail -3 $OUT/loop.err
exit 0
