#!/bin/bash
# the pyramid of a single-image call: one launch for all levels (k_pyramid_cascade, tile of the last level = ORBHIP_PC_TILE) against the seven launches (ORBHIP_PC_TILE=0)
TAG=${1:-cascade}
cd "$(dirname "$0")/.."
R=$(pwd); OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 100 python3 tools/single_frame_calls.py 2>&1 | grep single_frame | sed "s/^/$name /" >> $OUT/single.txt; }
for r in 1 2 3; do
  run levels ORBHIP_PC_TILE=0
  for t in 32x16 32x8 16x8 64x8 16x16 64x16; do run tile_$t ORBHIP_PC_TILE=$t; done
done
sort $OUT/single.txt
NROWS=7 tools/gpu_trace_variant.sh $TAG t32x8 ORBHIP_PC_TILE=32x8 | tail -8
