#!/bin/bash
# round 6, call Y: config 5 at random database sizes (the chunk rule of the balanced main pass) + the extractor / projection fuzzers with seeds no earlier call used
TAG=${1:-r06_y}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python tools/nn_size_fuzz.py 80 606 > $OUT/nn_size_fuzz_80cases.txt 2>&1
tail -2 $OUT/nn_size_fuzz_80cases.txt
timeout 1500 python tests/test_fuzz_gpu.py 300 606 > $OUT/gpu_fuzz_300cases_seed606.txt 2>&1
tail -1 $OUT/gpu_fuzz_300cases_seed606.txt
timeout 1500 python tests/test_parity_projection.py 300 606 > $OUT/gpu_projection_fuzz_300cases_seed606.txt 2>&1
tail -1 $OUT/gpu_projection_fuzz_300cases_seed606.txt
grep -c MISMATCH $OUT/*.txt
exit 0
