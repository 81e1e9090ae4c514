#!/bin/bash
# round 5, third call: stream-priority A/B (k_quadtree beside the blur), GPU tests, the driver's bench command, PMC passes at B = 512 and for the single-image
# kernels, rocprofv3 kernel stats + trace of the bench command, the secondary records of round 3 re-measured
TAG=${1:-r05_c}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
Q="--steps 20 --warmup 3 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop --no-secondary"
for rep in 1 2; do for m in 0 1 2; do ORBHIP_STREAM_PRIO=$m timeout 200 python bench.py $Q 2>> $OUT/prio.err | python -c "
import sys, json
b = json.loads(sys.stdin.readline()); print(json.dumps({'ORBHIP_STREAM_PRIO': $m, 'frames_per_s': b['value'], 'ms_per_step': b['ms_per_step'], 'repeats': b['repeats'], 'k_quadtree_ms': b['kernels_ms_per_launch']['k_quadtree'], 'k_blur_ms': b['kernels_ms_per_launch']['k_blur'], 'parity': b['parity']['mismatches']}))" >> $OUT/stream_prio_ab.jsonl; done; done
timeout 1500 python -m pytest tests -m gpu -x -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-io --no-traffic --no-dropin-loop --no-secondary > $OUT/rocprof_bench.json 2> $OUT/rocprof.err )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/bench_kernel_stats.csv; done
for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1); do python - "$f" > $OUT/quadtree_blur_overlap.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows]
ev.sort()
qt = [e for e in ev if e[2].startswith("k_quadtree")]; bl = [e for e in ev if e[2].startswith("k_blur")]
print("# per step: k_quadtree duration (us), k_blur duration (us), quadtree start - blur start (us), overlap (us)")
for q in qt:
    b = min(bl, key=lambda e: abs(e[0] - q[0]))
    ov = max(0, min(q[1], b[1]) - max(q[0], b[0]))
    print(f"{(q[1]-q[0])/1e3:8.1f} {(b[1]-b[0])/1e3:8.1f} {(q[0]-b[0])/1e3:8.1f} {ov/1e3:8.1f}")
PY
done
PMC_BATCH=512 bash tools/gpu_pmc.sh ${TAG}_pmc512 6 bench > $OUT/pmc512.log 2>&1
bash tools/gpu_pmc.sh ${TAG}_pmc_single 6 single > $OUT/pmc_single.log 2>&1
timeout 200 python tools/stereo_rate.py > $OUT/stereo_rate.json 2>> $OUT/bench.err
timeout 200 python tools/bow_rate.py > $OUT/bow_rate.json 2>> $OUT/bench.err
timeout 200 python tools/camera_rate.py > $OUT/camera_rate.json 2>> $OUT/bench.err
timeout 100 python tools/single_frame_calls.py > $OUT/single_frame.txt 2>&1
cat $OUT/stream_prio_ab.jsonl | cut -c1-330; grep -E "concurrency|local_mapping|sequence|passed|failed|error|exit" $OUT/pytest_gpu.log | tail -12; cut -c1-300 $OUT/bench.json; tail -4 $OUT/bench.err; head -14 $OUT/quadtree_blur_overlap.txt; cat $OUT/stereo_rate.json $OUT/bow_rate.json $OUT/camera_rate.json | cut -c1-600; tail -3 $OUT/single_frame.txt
