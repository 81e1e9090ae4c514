#!/bin/bash
TAG=${1:-bsw}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
for b in 512 1024 2048 512 1024 2048; do timeout 300 python3 bench.py --batch $b --steps 40 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io >> $OUT/sweep.jsonl 2>> $OUT/err.txt; done
python3 - <<PY
import json
for l in open("$OUT/sweep.jsonl"): d=json.loads(l); print(d["config"]["frames_per_step_per_gpu"], d["value"], d["ms_per_step"], d["parity"]["mismatches"], d["roofline"]["frac"])
PY
tail -2 $OUT/err.txt
