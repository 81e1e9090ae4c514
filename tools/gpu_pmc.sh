#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only) over a short bench; CSVs land in gpurun_out/$TAG/
# usage: tools/gpu_pmc.sh <tag> [number of passes, default all six]
TAG=${1:-pmc}; NPASS=${2:-7}; WHAT=${3:-bench}      # WHAT = bench (the B = $PMC_BATCH batch pipeline, default 256) | single (single-image calls + the stereo front-end loop) | r05 (bag of words, walks, FP4 scan)
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
CMD="python $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --batch ${PMC_BATCH:-256} --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop --no-secondary --parity-slots 0"
[ "$WHAT" = single ] && CMD="python $REPO/tools/pmc_single_workload.py"
[ "$WHAT" = r05 ] && CMD="python $REPO/tools/pmc_r05_workload.py"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  [ $i -gt $NPASS ] && break
  timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d /tmp/pmc_$TAG/p$i -o p$i -- $CMD > $OUT/p$i.stdout 2> $OUT/p$i.stderr
  f=$(find /tmp/pmc_$TAG/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/p${i}_counters.csv
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT
FETCH_SIZE TCC_HIT_sum
WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum
SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F6F4 SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
PASSES
ls -la $OUT | head -30
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/p*_counters.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[-1])
    for k, d in agg.items():
        print("  ", k, {c: round(v) for c, v in d.items()})
PY
