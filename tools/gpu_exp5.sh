#!/bin/bash
TAG=${1:-exp5}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_00_device.py tests/test_host_pipeline.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
timeout 600 python3 bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 tools/single_frame_calls.py > $OUT/single.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --sys-trace --output-format csv -d /tmp/strace -o st -- python3 $REPO/tools/single_frame_calls.py > $OUT/single_traced.txt 2>&1 )
python3 - > $OUT/single_timeline.txt <<'PY'
import csv, glob
def load(pat):
    f=glob.glob('/tmp/strace/**/*'+pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k=load('kernel_trace.csv'); m=load('memory_copy_trace.csv'); a=load('hip_api_trace.csv')
ev=[]
for r in k: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'K '+r['Kernel_Name'].split('(')[0]))
for r in m: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'M '+r.get('Direction','')+' '+r.get('Bytes',r.get('Size','?'))))
for r in a:
    if r['Function'] in ('hipStreamSynchronize','hipEventSynchronize','hipMemcpyAsync','hipLaunchKernel','hipModuleLaunchKernel','hipExtModuleLaunchKernel','hipMemcpy','hipEventRecord','hipStreamWaitEvent','hipPointerGetAttributes','hipSetDevice','hipGetLastError'):
        ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'A '+r['Function']))
ev.sort()
# last call = events after the last-but-one sync... take the last 60 events
last=ev[-70:]
t0=last[0][0]
for s,e,n in last: print(f'{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {n}')
PY
tail -3 $OUT/pytest.log; cat $OUT/single.txt; tail -1 $OUT/single_traced.txt
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['host_io'])"
tail -75 $OUT/single_timeline.txt
