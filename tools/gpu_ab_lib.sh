#!/bin/bash
# A/B of two builds of the library on one box: the bench's resident step (frames/s, per-kernel ms) alternating in-tree / $1, three rounds.   usage: tools/gpu_ab_lib.sh <other .so> [tag]
OTHER=$(realpath $1); TAG=${2:-ab_lib}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do
  for which in intree other; do
    L=""; [ $which = other ] && L=$OTHER
    ORBHIP_LIBRARY=$L timeout 300 python bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io --no-secondary --no-traffic --no-dropin-loop --parity-slots 4 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$which', d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_launch'].items() if v}, d['parity'].get('mismatches', d['parity']) if isinstance(d.get('parity'),dict) else d.get('parity'))" | cut -c1-400 | tee -a $OUT/ab.txt
  done
done
exit 0
