#!/bin/bash
# host-buffer path (pinned / pageable caller buffers, batch 256, ring) under the stream-priority settings, alternating; then the bench line without its long legs
TAG=${1:-r05_hostio2}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2 3; do for p in 0 2 1; do
  HOST_IO_ONLY=256 ORBHIP_STREAM_PRIO=$p timeout 200 python tools/host_io_rate.py 2>> $OUT/err.txt | grep summary | sed "s/^{/{\"ORBHIP_STREAM_PRIO\": $p, \"rep\": $rep, /" >> $OUT/host_io_prio.jsonl
done; done
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-dropin-loop > $OUT/bench_short.json 2>> $OUT/err.txt
cat $OUT/host_io_prio.jsonl | cut -c1-200; python - <<PY
import json
d=json.loads(open('$OUT/bench_short.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['host_io'])[:700])
PY
tail -2 $OUT/err.txt
exit 0
