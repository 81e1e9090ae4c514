#!/bin/bash
# round 6, call I: the whole -m gpu suite + smoke on the tree as it stands
TAG=${1:-r06_i}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -16 $OUT/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
exit 0
