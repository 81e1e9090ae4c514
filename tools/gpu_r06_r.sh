#!/bin/bash
# round 6, call R: where the hand-ordered superstep's time goes: text variants (ORBHIP_NN_BLOCK_VAR) and counters of the production text
TAG=${1:-r06_r}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in 0 1 2 3 0 1 2 3; do ORBHIP_NN_BLOCK_VAR=$v DB_EXPANDED=1 timeout 300 python tools/db_query_rate.py 2>&1 | tail -1 | cut -c1-200 | sed "s/^/var $v: /" | tee -a $OUT/rate.txt; done
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  DB_EXPANDED=1 timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d /tmp/pmc_$TAG/p$i -o p$i -- python $REPO/tools/db_query_rate.py > $OUT/p$i.stdout 2> $OUT/p$i.stderr
  f=$(find /tmp/pmc_$TAG/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/p${i}_counters.csv
done <<'PASSES'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
PASSES
DB_EXPANDED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_$TAG/kt -o kt -- python $REPO/tools/db_query_rate.py > $OUT/kt.stdout 2> $OUT/kt.stderr
cp $(find /tmp/pmc_$TAG/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/p*_counters.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print(f.split("/")[-1])
    for k, d in agg.items():
        if "fp4b" in k: print("  ", k, {c: round(v / n[(k, c)]) for c, v in d.items()}, "dispatches", max(n[(k, c)] for c in d))
PY
cut -c1-200 $OUT/kernel_stats.csv | head -8
exit 0
