#!/bin/bash
TAG=${1:-exp3}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
./tools/pcie_probe > $OUT/pcie_probe.txt 2>&1
HSA_ENABLE_SDMA=0 ./tools/pcie_probe > $OUT/pcie_probe_nosdma.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o kt -- python3 $REPO/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-host-io > $OUT/ktrace_bench.json 2> $OUT/ktrace.err )
f=$(find /tmp/ktrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/kernel_trace_summary.txt <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    name=r["Kernel_Name"].split("(")[0]; grid=r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    agg[(name,grid)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for (n,g),v in sorted(agg.items()):
    v=sorted(v); print(f"{n:32s} grid {g:>10s}  n {len(v):4d}  median {v[len(v)//2]:9.1f} us  min {v[0]:9.1f}")
# gaps on the timeline of the last step
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=rows[-40:]
print("--- last 40 dispatches: start(us rel), dur, name")
t0=int(last[0]["Start_Timestamp"])
for r in last: print(f'{(int(r["Start_Timestamp"])-t0)/1e3:10.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:9.1f}  {r["Kernel_Name"].split("(")[0]}  q{r.get("Queue_Id","?")}')
PY
cat $OUT/pcie_probe.txt $OUT/pcie_probe_nosdma.txt; cat $OUT/kernel_trace_summary.txt | head -80
