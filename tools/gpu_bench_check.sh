#!/bin/bash
# New bench.py paths: default run (torch-free, parity object), the N>1 control plane with one rank (library first, then torch + RCCL),
# two ranks sharing the GPU over gloo.   usage: <tag>
TAG=${1:-bchk}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
ORB_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; echo "exit $?" >> $OUT/bench_forcedist.err
ORB_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io > $OUT/bench_2rank_shared.json 2> $OUT/bench_2rank_shared.err; echo "exit $?" >> $OUT/bench_2rank_shared.err
head -8 $OUT/pytest.log; tail -4 $OUT/pytest.log; tail -3 $OUT/bench.err
python3 - <<PY
import json
for f in ("bench","bench_forcedist","bench_2rank_shared"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d.get("parity"), d.get("runtime"), d["roofline"]["frac"], d.get("host_io",{}).get("pinned"))
    except Exception as e: print(f,"failed",e); print(open("$OUT/%s.err"%f).read()[-1500:])
PY
