#!/bin/bash
# round 6, call B: the members and the loops again (quick), no test suite.   usage: tools/gpu_r06_b.sh <tag>
TAG=${1:-r06_b}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/secondary_units.py --blur-round-mode 1 --only matcher_calls > $OUT/secondary.json 2> $OUT/secondary.err
timeout 600 python tools/dropin_loop_rate.py kitti euroc > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
python - <<PY
import json
try:
    s = json.load(open("$OUT/secondary.json"))
    for k, v in s["matcher_calls"]["members"].items():
        print(f"{k[:70]:72s} gpu {v['gpu_ms']:.3f} (lib {v.get('gpu_ms_inside_the_library',0):.3f})  ref {v['ref_ms']:.3f} parity {v['parity']}")
    print(json.dumps(s["matcher_calls"]["back_end_loops"])[:600])
except Exception as e:
    print("secondary:", e); print(open("$OUT/secondary.err").read()[-1500:])
for l in open("$OUT/dropin_loop.jsonl"):
    d = json.loads(l); print(d["shape"][:30], d["ms_per_frame_gpu"], d["gpu_parts_ms"], d["parity"]["frames_mismatched"])
PY
tail -3 $OUT/loop.err
exit 0
