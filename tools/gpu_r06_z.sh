#!/bin/bash
# round 6, call Z: tiles per workgroup of k_pyramid_level_g again (ORBHIP_PYR_NT was chosen in round 3, before the loop overlapped its prefetch), B = 512 / 1024
TAG=${1:-r06_z}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
A="--steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io --no-secondary --no-dropin-loop --no-traffic --parity-slots 8"
for r in 1 2 3; do
  for nt in 4 2 8 3 6; do ORBHIP_PYR_NT=$nt timeout 300 python bench.py $A 2>> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt $nt', d['value'], d['kernels_ms_per_launch']['k_pyramid_level'], d['parity']['mismatches'])" | tee -a $OUT/nt.txt; done
done
for b in 512 1024 2048; do timeout 600 python bench.py $A --batch $b 2>> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', d['value'], d['ms_per_step'], d['kernels_ms_per_launch'], d['parity']['mismatches'])" | tee -a $OUT/batch.txt; done
tail -3 $OUT/err.txt
exit 0
