#!/bin/bash
# round 6, call E2: bench.py's own in-process host_io figure under variations of what the process did before
TAG=${1:-r06_e2}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F="--no-traffic --no-dropin-loop --no-secondary --no-cpu-baseline"
run() { name=$1; shift; "$@" > $OUT/$name.json 2>> $OUT/err.txt; python - <<PY
import json
d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
h = d["host_io"]
print("$name", "fresh", h["pinned"]["frames_per_s"], h["pageable"]["frames_per_s"], "in-process", h["in_this_process"])
PY
}
run default python bench.py $F
run short python bench.py $F --steps 5 --repeats 1
run noparity python bench.py $F --parity-slots 0
run prio0 env ORBHIP_STREAM_PRIO=0 python bench.py $F
run batch64 python bench.py $F --batch 64
tail -3 $OUT/err.txt
exit 0
