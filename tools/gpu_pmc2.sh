#!/bin/bash
# two PMC passes in serial mode (each kernel alone): stall composition and MFMA / TA activity per kernel.  usage: tools/gpu_pmc2.sh <tag>
TAG=${1:-pmc2}; OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --batch 256 --no-cpu-baseline --no-host-io"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  ORBHIP_SERIAL=1 timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d /tmp/pmc_$TAG/p$i -o p$i -- $CMD > $OUT/p$i.stdout 2> $OUT/p$i.stderr
  f=$(find /tmp/pmc_$TAG/p$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/p${i}_counters.csv
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
PASSES
cd $REPO
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/p*_counters.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[-1])
    for k, d in agg.items():
        if k.startswith("k_") and not k.startswith("k_match"): print("  ", k, {c: round(v / max(d.get("SQ_WAVES", 1), 1), 1) if c.startswith("SQ_") and c != "SQ_WAVES" and "SQ_WAVES" in d else round(v) for c, v in d.items()})
PY
grep -i "error\|invalid" $OUT/p*.stderr | head -5
