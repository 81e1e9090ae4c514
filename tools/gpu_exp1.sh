#!/bin/bash
TAG=${1:-exp1}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
timeout 600 python3 bench.py --pool --steps 20 --warmup 2 --repeats 3 > $OUT/bench_pool.json 2> $OUT/bench_pool.err; echo "exit $?" >> $OUT/bench_pool.err
for PSV in 0 96; do
  ORBHIP_FC_PSTRIDE=$PSV timeout 300 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io > $OUT/bench_ps$PSV.json 2>> $OUT/bench.err
done
ORB_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io > $OUT/bench_2rank_shared.json 2> $OUT/bench_2rank_shared.err; echo "exit $?" >> $OUT/bench_2rank_shared.err
head -6 $OUT/pytest.log; tail -5 $OUT/pytest.log
python3 - <<PY
import json
for f in ("bench_pool","bench_ps0","bench_ps96","bench_2rank_shared"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d.get("parity",{}).get("mismatches"), d.get("kernels_ms_per_launch"), d["runtime"]["mapped"], d["runtime"].get("control_plane"))
    except Exception as e: print(f,"failed",e)
PY
tail -3 $OUT/bench_pool.err $OUT/bench_2rank_shared.err $OUT/bench.err
