#!/bin/bash
# round 6: the long fuzz campaign on the round's last library - extractor + SearchForInitialization, the projection searches, the conflict-heavy matcher cases, config 5 sizes - seeds no earlier call used
TAG=${1:-r06_fz}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python tests/test_fuzz_gpu.py 2000 2026 > $OUT/gpu_fuzz_2000cases.txt 2>&1; tail -1 $OUT/gpu_fuzz_2000cases.txt
timeout 2400 python tests/test_parity_projection.py 2000 2026 > $OUT/gpu_projection_fuzz_2000cases.txt 2>&1; tail -1 $OUT/gpu_projection_fuzz_2000cases.txt
timeout 1800 python tests/test_fuzz_matchers.py $REPO/orb_slam2_amd/liborbhip.so 1000 > $OUT/gpu_matcher_conflict_fuzz_2000cases.txt 2>&1; tail -1 $OUT/gpu_matcher_conflict_fuzz_2000cases.txt
timeout 1800 python tools/nn_size_fuzz.py 300 2026 > $OUT/nn_size_fuzz_300cases.txt 2>&1; tail -1 $OUT/nn_size_fuzz_300cases.txt
grep -c "MISMATCH\|FAIL\|differ" $OUT/*.txt
exit 0
