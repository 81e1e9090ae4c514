#!/bin/bash
# k_stereo_rows on 1024 threads: stereo tests, loops, a traced KITTI loop
TAG=${1:-r05_r}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_stereo.py tests/test_dropin_loop.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 600 python tools/dropin_loop_rate.py kitti euroc > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
timeout 200 python tools/stereo_rate.py > $OUT/stereo_rate.json 2>> $OUT/loop.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_loop_$TAG -o loop -- python $REPO/tools/dropin_loop_rate.py kitti > /dev/null 2>> $OUT/loop.err )
for f in $(find /tmp/prof_loop_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/loop_kernel_stats.csv; done
tail -2 $OUT/pytest_gpu.log; python - <<PY
import json,csv
for l in open('$OUT/dropin_loop.jsonl'):
    d=json.loads(l); print(d['shape'][:18], d['ms_per_frame_gpu'], d['gpu_parts_ms'], d['parity']['frames_mismatched'])
print(open('$OUT/stereo_rate.json').read()[:200])
for r in csv.DictReader(open('$OUT/loop_kernel_stats.csv')):
    if 'stereo' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'])
PY
exit 0
