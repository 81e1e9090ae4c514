#!/bin/bash
TAG=${1:-r06_e3}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F="--no-traffic --no-dropin-loop --no-secondary --no-cpu-baseline --steps 5 --repeats 1 --parity-slots 0"
run() { name=$1; shift; "$@" > $OUT/$name.json 2>> $OUT/err.txt; python - <<PY
import json
d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
h = d["host_io"]
print("$name", "fresh", h["pinned"]["frames_per_s"], h["pageable"]["frames_per_s"], "in-process", h["in_this_process"], "value", d["value"])
PY
}
run q8 env GPU_MAX_HW_QUEUES=8 python bench.py $F
run q2 env GPU_MAX_HW_QUEUES=2 python bench.py $F
run q16 env GPU_MAX_HW_QUEUES=16 python bench.py $F
run prio1 env ORBHIP_STREAM_PRIO=1 python bench.py $F
tail -3 $OUT/err.txt
exit 0
