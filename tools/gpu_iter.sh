#!/bin/bash
# One kernel-tuning iteration on the GPU: extraction parity tests, standalone (serial) kernel times at B = 256, the production
# schedule at B = 512, VALU / LDS instructions per wave.  usage: tools/gpu_iter.sh <tag> [extra command run first]
TAG=${1:-it}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
[ -n "$2" ] && bash -c "$2" > $OUT/extra.log 2>&1
timeout 600 python -m pytest tests/test_parity_extract.py tests/test_full_size_gpu.py tests/test_reference_extractor.py tests/test_golden_frames.py tests/test_parity_match.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
ORBHIP_SERIAL=1 timeout 300 python bench.py --steps 20 --warmup 3 --repeats 3 --batch 256 --no-cpu-baseline --no-host-io > $OUT/bench_serial_b256.json 2>> $OUT/bench.err
timeout 300 python bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io > $OUT/bench_b512.json 2>> $OUT/bench.err
REPO=$(pwd); cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d /tmp/pmc_$TAG/p1 -o p1 -- python $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --batch 256 --no-cpu-baseline --no-host-io > $REPO/$OUT/p1.stdout 2> $REPO/$OUT/p1.stderr
f=$(find /tmp/pmc_$TAG/p1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$OUT/p1_counters.csv
cd $REPO
[ -f $OUT/extra.log ] && tail -20 $OUT/extra.log
python3 - <<PY
import csv, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
try:
    for r in csv.DictReader(open("$OUT/p1_counters.csv")):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        if d.get("SQ_WAVES") and k.startswith("k_"): print(k, "VALU/wave", round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1), "LDS/wave", round(d["SQ_INSTS_LDS"] / d["SQ_WAVES"], 1), "SALU/wave", round(d["SQ_INSTS_SALU"] / d["SQ_WAVES"], 1), "waves", int(d["SQ_WAVES"]))
except Exception as e: print("pmc parse failed", e)
for f in ("bench_serial_b256", "bench_b512"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_launch"].items() if v}, d["roofline"]["frac"])
    except Exception as e: print(f, "failed", e)
PY
tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err
