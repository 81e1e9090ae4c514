#!/bin/bash
# round 6, last session: the fuzzers on the round's last library with seeds no earlier call used - FuseBatch's shared / held entries against the single-slot entry,
# extractor + SearchForInitialization, the projection searches, the conflict-heavy matcher cases, config 5 sizes
TAG=${1:-r06_fs}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python tools/fuse_shared_fuzz.py 1500 4242 > $OUT/gpu_fuse_shared_held_fuzz_1500cases.txt 2>&1; tail -1 $OUT/gpu_fuse_shared_held_fuzz_1500cases.txt
timeout 1200 python tests/test_fuzz_gpu.py 600 9090 > $OUT/gpu_fuzz_600cases_seed9090.txt 2>&1; tail -1 $OUT/gpu_fuzz_600cases_seed9090.txt
timeout 1200 python tests/test_parity_projection.py 600 9090 > $OUT/gpu_projection_fuzz_600cases_seed9090.txt 2>&1; tail -1 $OUT/gpu_projection_fuzz_600cases_seed9090.txt
timeout 900 python tests/test_fuzz_matchers.py $REPO/orb_slam2_amd/liborbhip.so 300 > $OUT/gpu_matcher_conflict_fuzz_seed_default.txt 2>&1; tail -1 $OUT/gpu_matcher_conflict_fuzz_seed_default.txt
timeout 900 python tools/nn_size_fuzz.py 100 9090 > $OUT/nn_size_fuzz_100cases_seed9090.txt 2>&1; tail -1 $OUT/nn_size_fuzz_100cases_seed9090.txt
grep -c "MISMATCH\|FAIL\|differ" $OUT/*.txt
exit 0
