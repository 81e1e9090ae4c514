#!/bin/bash
# Short gpurun call: GPU parity tests, then bench at B = 64 / 256 / 512 (no CPU baseline).  usage: tools/gpu_quick.sh <tag>
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
for cfg in "64 1" "256 1" "512 1"; do set -- $cfg; timeout 200 python bench.py --steps 20 --warmup 3 --batch $1 --streams $2 --no-cpu-baseline >> $OUT/bench_sweep.jsonl 2>> $OUT/bench.err; done
tail -4 $OUT/pytest_gpu.log
python - <<PY
import json
for l in open('$OUT/bench_sweep.jsonl'):
    d=json.loads(l); print(d['config']['frames_per_step_per_gpu'], d['config']['streams_per_gpu'], d['value'], d['kernels_ms_per_launch'])
PY
tail -3 $OUT/bench.err
