#!/bin/bash
TAG=${1:-exp4}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "not config5 and not config4" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
timeout 300 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io > $OUT/bench.json 2>> $OUT/bench.err
for b in 64 128 256; do timeout 300 python3 bench.py --batch $b --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io >> $OUT/bench_sweep.jsonl 2>> $OUT/bench.err; done
tail -4 $OUT/pytest.log
python3 - <<PY
import json
for l in [open("$OUT/bench.json").read().strip().splitlines()[-1]]+open("$OUT/bench_sweep.jsonl").read().strip().splitlines():
    d=json.loads(l); print(d["config"]["frames_per_step_per_gpu"], d["value"], d["ms_per_step"], d.get("parity",{}).get("mismatches"), {k:v for k,v in d["kernels_ms_per_launch"].items() if v})
PY
tail -3 $OUT/bench.err
