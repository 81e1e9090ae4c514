#!/bin/bash
# round 6, call G: pyramid tile loop without vmcnt(0) waits - parity first, then the level launches' time
TAG=${1:-r06_g}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_extract.py tests/test_pyramid_cascade.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -1 $OUT/pytest.txt
F="--no-traffic --no-dropin-loop --no-secondary --no-cpu-baseline --no-host-io --steps 20 --repeats 2 --parity-slots 16"
run() { name=$1; shift; timeout 600 "$@" > $OUT/$name.json 2>> $OUT/err.txt; python - <<PY
import json
d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["kernels_ms_per_launch"]["k_pyramid_level"], d["kernels_ms_per_launch"]["k_describe"], d["parity"]["mismatches"], d["parity"]["replica_mismatches"])
PY
}
run base python bench.py $F


run again python bench.py $F
tail -3 $OUT/err.txt
exit 0
