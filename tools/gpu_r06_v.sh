#!/bin/bash
# round 6, call V: counters of the hand-ordered superstep, production text against ORBHIP_NN_BLOCK_VAR=64 (kept pairs ignored)
TAG=${1:-r06_v}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in 0 64; do
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  ORBHIP_NN_BLOCK_VAR=$v DB_EXPANDED=1 timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d /tmp/pmc_$TAG/v${v}p$i -o p$i -- python $REPO/tools/db_query_rate.py > $OUT/v${v}p$i.stdout 2> $OUT/v${v}p$i.stderr
  f=$(find /tmp/pmc_$TAG/v${v}p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/v${v}p${i}_counters.csv
done <<'PASSES'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR
PASSES
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/v*_counters.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print(f.split("/")[-1])
    for k, d in agg.items():
        if "fp4b<15, true" in k: print("  ", k, {c: round(v / n[(k, c)]) for c, v in d.items()}, "dispatches", max(n[(k, c)] for c in d))
PY
exit 0
