#!/bin/bash
# bench at several (batch, streams) points.  usage: tools/gpu_streams.sh <tag> "B S" "B S" ...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
for cfg in "$@"; do set -- $cfg; timeout 200 python bench.py --steps 20 --warmup 3 --batch $1 --streams $2 --no-cpu-baseline >> $OUT/sweep.jsonl 2>> $OUT/bench.err; done
python - <<PY
import json
for l in open('$OUT/sweep.jsonl'):
    d=json.loads(l); print(d['config']['frames_per_step_per_gpu'], d['config']['streams_per_gpu'], d['value'], d['ms_per_step'], d['kernels_ms_per_launch'])
PY
