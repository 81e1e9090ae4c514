#!/bin/bash
# round 5: where k_bow_assemble's time goes (kernel cut short at successive points, PROFILING build) + the 16-lanes-per-feature descent
TAG=${1:-r05_j}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bow.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
python tools/bow_single.py 1000 200 > $OUT/single.txt 2>&1
python tools/bow_single.py 2000 200 >> $OUT/single.txt 2>&1
for n in 1000 2000; do
for st in 10 20 25 30 40 50 60 70 0; do
  ( cd /tmp && ORBHIP_BOW_STOP=$st timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bs_${n}_$st -o p -- python $REPO/tools/bow_single.py $n 100 > /dev/null 2>> $OUT/err.txt )
  f=$(find /tmp/prof_bs_${n}_$st -name "*kernel_stats.csv" | head -1)
  echo "n=$n stop=$st $(grep -E 'k_bow_assemble|k_bow_descend' $f | cut -d, -f1-4 | tr '\n' ' ')" >> $OUT/stops.txt
done; done
timeout 300 python tools/bow_rate.py > $OUT/bow_rate.json 2>> $OUT/err.txt
tail -3 $OUT/pytest_gpu.log; cat $OUT/single.txt $OUT/stops.txt; cut -c1-400 $OUT/bow_rate.json; tail -3 $OUT/err.txt
exit 0
