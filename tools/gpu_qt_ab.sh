#!/bin/bash
# quadtree phase stamps (ab/liborbhip_qttrace.so), then the in-tree library against ab/liborbhip_prev.so: single-image call, B = 512 bench, standalone kernel times (ORBHIP_SERIAL=1)
cd "$(dirname "$0")/.."
R=$(pwd)
ORBHIP_LIBRARY=$R/ab/liborbhip_qttrace.so python3 tools/qt_trace_experiment.py 2>&1 | grep -A1 "1241x376 level [037]:\|752x480 level 0" | grep -v "^--" | cut -c1-330
for r in 1 2; do
  ORBHIP_LIBRARY=$R/ab/liborbhip_prev.so timeout 100 python3 tools/single_frame_calls.py 2>&1 | grep single_frame | sed 's/^/prev /'
  timeout 100 python3 tools/single_frame_calls.py 2>&1 | grep single_frame | sed 's/^/new  /'
done
for n in new prev; do
  [ $n = prev ] && export ORBHIP_LIBRARY=$R/ab/liborbhip_prev.so || unset ORBHIP_LIBRARY
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', j['value'], j['parity']['mismatches'], {k: round(v,4) for k,v in j['kernels_ms_per_launch'].items() if v})"
  ORBHIP_SERIAL=1 timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n serial', j['value'], {k: round(v,4) for k,v in j['kernels_ms_per_launch'].items() if v})"
done
