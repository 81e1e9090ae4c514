#!/bin/bash
# round 6, call F: pyramid at batch - the level launches vs the cascade (and tile shapes of the cascade)
TAG=${1:-r06_f}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F="--no-traffic --no-dropin-loop --no-secondary --no-cpu-baseline --no-host-io --steps 40 --repeats 3 --parity-slots 8"
run() { name=$1; shift; "$@" > $OUT/$name.json 2>> $OUT/err.txt; python - <<PY
import json
d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v > 0.01}, d["parity"]["mismatches"], d.get("INVALID"))
PY
}
run levels python bench.py $F
run cascade env ORBHIP_PC_ALWAYS=1 python bench.py $F
run cascade_64x16 env ORBHIP_PC_ALWAYS=1 ORBHIP_PC_TILE=64x16 python bench.py $F
run cascade_32x32 env ORBHIP_PC_ALWAYS=1 ORBHIP_PC_TILE=32x32 python bench.py $F
run serial_levels env ORBHIP_SERIAL=1 python bench.py $F
tail -3 $OUT/err.txt
exit 0
