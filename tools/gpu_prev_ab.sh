#!/bin/bash
# the in-tree library against ab/liborbhip_prev.so (the previous commit's kernels), interleaved: single-image call, front-end loop, the B = 512 bench; then the GPU suite
TAG=${1:-prevab}
cd "$(dirname "$0")/.."
R=$(pwd); OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for r in 1 2 3; do
  ORBHIP_LIBRARY=$R/ab/liborbhip_prev.so timeout 100 python3 tools/single_frame_calls.py 2>&1 | grep single_frame | sed 's/^/prev /' >> $OUT/single.txt
  timeout 100 python3 tools/single_frame_calls.py 2>&1 | grep single_frame | sed 's/^/new  /' >> $OUT/single.txt
done
sort $OUT/single.txt
NROWS=7 tools/gpu_trace_variant.sh $TAG new | grep -v "^W2026" | tail -8
for r in 1 2; do
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop > $OUT/bench_new_$r.json 2>> $OUT/bench.err
  ORBHIP_LIBRARY=$R/ab/liborbhip_prev.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop > $OUT/bench_prev_$r.json 2>> $OUT/bench.err
done
python3 - <<PY
import json
for n in ("new_1","prev_1","new_2","prev_2"):
    try:
        j = json.loads(open("$OUT/bench_%s.json" % n).read().strip().splitlines()[-1]); print(n, j["value"], j["ms_per_step"], j["parity"]["mismatches"], {k: round(v, 4) for k, v in j.get("kernels_ms_per_launch", {}).items() if v})
    except Exception as e: print(n, "failed", e)
PY
timeout 200 python3 tools/dropin_loop_rate.py 2>/dev/null > $OUT/loop_new.jsonl
python3 - <<PY
import json
for l in open("$OUT/loop_new.jsonl"):
    js = json.loads(l); print(js["shape"][:9], js["ms_per_frame_gpu"], js["gpu_parts_ms"], js["parity"]["frames_mismatched"])
PY
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
