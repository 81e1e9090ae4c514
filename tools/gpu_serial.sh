#!/bin/bash
# Standalone kernel times: ORBHIP_SERIAL=1 puts every kernel on one stream (no overlap).  usage: tools/gpu_serial.sh <tag>
TAG=${1:-ser}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
ORBHIP_SERIAL=1 timeout 200 python bench.py --steps 20 --warmup 3 --batch 256 --no-cpu-baseline >> $OUT/bench_serial.jsonl 2>> $OUT/bench.err
timeout 200 python bench.py --steps 20 --warmup 3 --batch 256 --no-cpu-baseline >> $OUT/bench_serial.jsonl 2>> $OUT/bench.err
timeout 200 python bench.py --steps 20 --warmup 3 --batch 256 --no-cpu-baseline --extract-only >> $OUT/bench_serial.jsonl 2>> $OUT/bench.err
ORBHIP_SERIAL=1 timeout 200 python bench.py --steps 20 --warmup 3 --batch 256 --no-cpu-baseline --extract-only >> $OUT/bench_serial.jsonl 2>> $OUT/bench.err
timeout 200 python bench.py --steps 20 --warmup 3 --batch 512 --no-cpu-baseline >> $OUT/bench_serial.jsonl 2>> $OUT/bench.err
python - <<PY
import json
for l in open('$OUT/bench_serial.jsonl'):
    d=json.loads(l); print(d['metric'][:12], d['config']['frames_per_step_per_gpu'], d['value'], d['ms_per_step'], d['kernels_ms_per_launch'], round(sum(d['kernels_ms_per_launch'].values()),3))
PY
