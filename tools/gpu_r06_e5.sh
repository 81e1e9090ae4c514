#!/bin/bash
TAG=${1:-r06_e5}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F="--no-traffic --no-dropin-loop --no-secondary --no-cpu-baseline --steps 5 --repeats 1 --parity-slots 0"
run() { name=$1; shift; "$@" > $OUT/$name.json 2>> $OUT/err.txt; python - <<PY
import json
d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
h = d["host_io"]
print("$name", "fresh", h["pinned"]["frames_per_s"], h["pageable"]["frames_per_s"], "in-process", h["in_this_process"])
PY
}
for pp in hl lh ll hh nl ln; do run copy_$pp env ORBHIP_COPY_STREAM_PRIO=$pp python bench.py $F; done
for pp in hl ll; do run q2_copy_$pp env GPU_MAX_HW_QUEUES=2 ORBHIP_COPY_STREAM_PRIO=$pp python bench.py $F; done
tail -3 $OUT/err.txt
exit 0
