#!/bin/bash
# round 5: where k_bow_assemble's time goes (kernel cut short at successive points, PROFILING build) + the 16-lanes-per-feature descent
TAG=${1:-r05_j}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python tools/bow_single.py 1000 200 > $OUT/single.txt 2>&1
python tools/bow_single.py 2000 200 >> $OUT/single.txt 2>&1
for n in 1000 2000; do
for st in 10 20 25 30 40 50 60 70 0; do
  ( cd /tmp && ORBHIP_BOW_STOP=$st timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bs_${n}_$st -o p -- python $REPO/tools/bow_single.py $n 100 > /dev/null 2>> $OUT/err.txt )
  f=$(find /tmp/prof_bs_${n}_$st -name "*kernel_stats.csv" | head -1)
  python - "$f" $n $st >> $OUT/stops.txt <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "k_bow" in r["Name"]}
print("n=%s stop=%s " % (sys.argv[2], sys.argv[3]) + "  ".join("%s %.2f us" % (k.replace("void ", ""), v) for k, v in sorted(rows.items())))
PY
done; done
cat $OUT/single.txt $OUT/stops.txt; tail -3 $OUT/err.txt
exit 0
