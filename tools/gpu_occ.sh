#!/bin/bash
TAG=${1:-occ}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
for pad in 0 6000 11000 19000 32000 0; do
  ORBHIP_FC_PAD_LDS=$pad ORBHIP_SERIAL=1 timeout 300 python3 bench.py --steps 30 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io --parity-slots 0 > $OUT/b.json 2>> $OUT/err.txt
  python3 -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('pad', $pad, 'fast', d['kernels_ms_per_launch']['k_fast_cells'], 'value', d['value'])"
done
tail -2 $OUT/err.txt
