#!/bin/bash
TAG=${1:-r06_e4}
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
run probe python bench.py $F
run probe_q2 env GPU_MAX_HW_QUEUES=2 python bench.py $F
run probe_prio1 env ORBHIP_STREAM_PRIO=1 python bench.py $F
run noprobe env ORBHIP_COPY_STREAM_PROBE=0 python bench.py $F
timeout 600 python -m pytest tests/test_host_pipeline.py tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -2
tail -3 $OUT/err.txt
exit 0
