#!/bin/bash
# schedules x library variants in one call    usage: tools/gpu_sched.sh <tag> "<variants>" "<scheds>"
TAG=${1:-sch}; VARS=${2:-new}; SCHEDS=${3:-"0 4"}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for r in 1 2; do
  for v in $VARS; do for sc in $SCHEDS; do
    if [ $v = new ]; then unset ORBHIP_LIBRARY; else export ORBHIP_LIBRARY=$REPO/ab/liborbhip_$v.so; fi
    ORBHIP_SCHED=$sc timeout 300 python3 bench.py --steps 40 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io >> $OUT/bench_${v}_s$sc.jsonl 2>> $OUT/err.txt
  done; done
done
unset ORBHIP_LIBRARY
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.jsonl")):
    rows=[json.loads(l) for l in open(f).read().strip().splitlines()]
    ks=[k for k in rows[0]["kernels_ms_per_launch"] if rows[0]["kernels_ms_per_launch"][k]]
    print(f.split("/")[-1], [r["value"] for r in rows], "parity", [r["parity"]["mismatches"] for r in rows], {k: round(sum(r["kernels_ms_per_launch"][k] for r in rows)/len(rows),4) for k in ks})
PY
tail -2 $OUT/err.txt
