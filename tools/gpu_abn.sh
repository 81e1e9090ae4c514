#!/bin/bash
# N-way A/B in one call (boxes differ by a few per cent; rounds are interleaved).
# usage: tools/gpu_abn.sh <tag> <rounds> "<label>|<lib name under ab/ or ->|<ENV=val ...>|<bench args>" ...
TAG=${1:-abn}; ROUNDS=${2:-3}; shift; shift
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for round in $(seq $ROUNDS); do
  for spec in "$@"; do
    IFS='|' read -r label lib envs args <<< "$spec"
    ( [ "$lib" != "-" ] && export ORBHIP_LIBRARY=$(pwd)/ab/liborbhip_$lib.so
      for e in $envs; do export $e; done
      timeout 300 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io $args >> $OUT/bench_$label.jsonl 2>> $OUT/bench.err )
  done
done
python3 - "$OUT" "$@" <<'PY'
import json, sys
out = sys.argv[1]
for spec in sys.argv[2:]:
    label = spec.split('|')[0]
    try:
        rows = [json.loads(l) for l in open("%s/bench_%s.jsonl" % (out, label)).read().strip().splitlines()]
    except Exception as e:
        print(label, "failed", e); continue
    ks = rows[0]["kernels_ms_per_launch"].keys()
    print(label, [r["value"] for r in rows], "parity", [r["parity"]["mismatches"] for r in rows])
    print("   ", {k.replace("k_", ""): round(sum(r["kernels_ms_per_launch"][k] for r in rows) / len(rows), 4) for k in ks if rows[0]["kernels_ms_per_launch"][k]})
PY
tail -2 $OUT/bench.err
