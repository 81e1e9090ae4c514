#!/bin/bash
# round 6, the record call on the final tree (the hand-ordered FP4 scan with rows per workgroup sized by the database): every GPU test, smoke, the driver's bench command, rocprofv3 kernel stats of the same command, the secondary tools,
# the front-end loops (stereo KITTI / EuRoC, mono, RGB-D) plain and traced, the N > 1 path on one GPU, the fuzz cases.   usage: tools/gpu_r06_final.sh <tag>
TAG=${1:-r06_final4}
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
SECONDS=0
timeout 1200 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? after $SECONDS s" >> $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python $REPO/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io --no-secondary > $OUT/rocprof_bench.json 2> $OUT/rocprof.err )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; done
timeout 600 python tools/dropin_loop_rate.py kitti euroc mono rgbd > $OUT/dropin_loop.jsonl 2> $OUT/loop.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_loop_$TAG -o loop -- python $REPO/tools/dropin_loop_rate.py kitti > /dev/null 2>> $OUT/loop.err )
for k in kernel_stats kernel_trace memory_copy_trace; do for f in $(find /tmp/prof_loop_$TAG -name "*${k}.csv" | head -1); do cp $f $OUT/loop_$k.csv; done; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_mono_$TAG -o loop -- python $REPO/tools/dropin_loop_rate.py mono > /dev/null 2>> $OUT/loop.err )
for k in kernel_stats kernel_trace memory_copy_trace; do for f in $(find /tmp/prof_mono_$TAG -name "*${k}.csv" | head -1); do cp $f $OUT/mono_$k.csv; done; done
timeout 200 python tools/stereo_rate.py > $OUT/stereo_rate.json 2>> $OUT/tools.err
timeout 200 python tools/matcher_latency.py > $OUT/matcher_latency.json 2>> $OUT/tools.err
timeout 300 python tools/bow_rate.py > $OUT/bow_rate.json 2>> $OUT/tools.err
timeout 300 python tools/camera_rate.py > $OUT/camera_rate.json 2>> $OUT/tools.err
timeout 120 python tools/host_io_rate.py > $OUT/host_io.jsonl 2>> $OUT/tools.err
timeout 120 python tools/db_query_rate.py > $OUT/db_query_rate.jsonl 2>> $OUT/tools.err
DB_EXPANDED=1 timeout 120 python tools/db_query_rate.py >> $OUT/db_query_rate.jsonl 2>> $OUT/tools.err
ORBHIP_NN_STATS=1 DB_EXPANDED=1 timeout 120 python tools/db_query_rate.py 2>&1 | grep "kept" | tail -1 >> $OUT/db_query_rate.jsonl
timeout 900 python tools/db_full_parity.py > $OUT/db_full_parity.json 2>> $OUT/tools.err
( cd /tmp; i=0
  for line in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    DB_EXPANDED=1 timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d /tmp/pmc_nn_$TAG/p$i -o p$i -- python $REPO/tools/db_query_rate.py > /dev/null 2>> $OUT/tools.err
    f=$(find /tmp/pmc_nn_$TAG/p$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/nn_pmc_p${i}_counters.csv
  done
  DB_EXPANDED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_nn_$TAG/kt -o kt -- python $REPO/tools/db_query_rate.py > /dev/null 2>> $OUT/tools.err
  cp $(find /tmp/pmc_nn_$TAG/kt -name "*kernel_stats.csv" | head -1) $OUT/nn_kernel_stats.csv )
python3 - > $OUT/nn_pmc_summary.txt <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/nn_pmc_p*_counters.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "fp4b" in k: print(k, {c: round(v / n[(k, c)]) for c, v in d.items()}, "dispatches", max(n[(k, c)] for c in d))
PY
timeout 120 python tools/bow_single.py 1000 200 > $OUT/bow_single.txt 2>> $OUT/tools.err; timeout 120 python tools/bow_single.py 2000 200 >> $OUT/bow_single.txt 2>> $OUT/tools.err
ORB_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --steps 50 --warmup 3 --repeats 3 --batch 256 --no-cpu-baseline --no-host-io --no-secondary > $OUT/bench_2rank_shared.json 2>> $OUT/tools.err
timeout 300 python bench.py --pool --steps 20 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_pool.json 2>> $OUT/tools.err
timeout 600 python tests/test_fuzz_gpu.py 120 78 > $OUT/gpu_fuzz_120cases.txt 2>&1
for n in 100 1000 3000; do DB_EXPANDED=1 timeout 120 python tools/db_query_rate.py $n >> $OUT/db_query_rate_sizes.jsonl 2>> $OUT/tools.err; done
tail -4 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; cut -c1-1200 $OUT/bench.json; tail -2 $OUT/bench.err; head -8 $OUT/kernel_stats.csv | cut -d, -f1-4
cut -c1-500 $OUT/dropin_loop.jsonl; cut -c1-300 $OUT/bench_2rank_shared.json; cut -c1-300 $OUT/bench_pool.json; tail -2 $OUT/gpu_fuzz_120cases.txt; tail -3 $OUT/tools.err; cat $OUT/nn_pmc_summary.txt | cut -c1-500; cut -c1-300 $OUT/db_query_rate.jsonl
exit 0
