#!/bin/bash
# kernel timeline of single-frame calls (tools/single_frame_calls.py) under rocprofv3 --kernel-trace.  usage: gpu_trace_variant.sh <tag> <name> [ENV=VALUE ...]
TAG=$1; NAME=$2; shift; shift
cd "$(dirname "$0")/.."
R=$(pwd); OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG$NAME -o s -- python3 $R/tools/single_frame_calls.py > $R/$OUT/$NAME.run.txt 2>&1 )
for f in $(find /tmp/prof_$TAG$NAME -name "*kernel_trace.csv" | head -1); do python3 - $f > $OUT/$NAME.trace.txt <<PY
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))[-${NROWS:-24}:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows: print("%9.2f us + %7.2f  -> %9.2f  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:36]))
PY
done
for f in $(find /tmp/prof_$TAG$NAME -name "*kernel_stats.csv" | head -1); do cp $f $OUT/$NAME.kernel_stats.csv; done
grep single_frame $OUT/$NAME.run.txt; cat $OUT/$NAME.trace.txt
