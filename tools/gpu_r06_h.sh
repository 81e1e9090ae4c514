#!/bin/bash
# round 6, call H: select kernels with their tables in device memory (nfeatures 12000 at 1080p) + the matcher suites
TAG=${1:-r06_h}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_large_feature_counts.py tests/test_parity_match.py tests/test_parity_projection.py -m gpu -x -q --durations=8 > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest.txt
exit 0
