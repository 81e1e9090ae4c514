#!/bin/bash
# One call: GPU parity suite on the in-tree build, then A/B of ab/liborbhip_<name>.so against it (interleaved, 3 rounds).   usage: tools/gpu_ab_tests.sh <tag> <name> [bench args]
TAG=${1:-abt}; NAME=${2:-base}; shift; shift
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
bash tools/gpu_ab.sh $TAG $NAME "$@"
