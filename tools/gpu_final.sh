#!/bin/bash
# The round's measurement call: every GPU test, smoke, full-size config-5 parity, the default bench (host_io + cpu_baseline), rocprofv3
# kernel stats of the same command, batch sweep, serial (standalone) kernel times, schedule variants, PMC passes, secondary tools.
# usage: tools/gpu_final.sh <tag>
TAG=${1:-fin}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
REPO=$(pwd)
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $OUT/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o orb -- python $REPO/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-host-io > $REPO/$OUT/rocprof_bench.json 2> $REPO/$OUT/rocprof.err )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; done
for b in 64 128 256 512 1024; do timeout 200 python bench.py --steps 30 --warmup 3 --repeats 3 --batch $b --no-cpu-baseline --no-host-io >> $OUT/bench_sweep.jsonl 2>> $OUT/bench.err; done
ORBHIP_SERIAL=1 timeout 200 python bench.py --steps 20 --warmup 3 --repeats 3 --batch 256 --no-cpu-baseline --no-host-io > $OUT/bench_serial_b256.json 2>> $OUT/bench.err
ORBHIP_BLUR=valu timeout 200 python bench.py --steps 50 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-io > $OUT/bench_blur_valu.json 2>> $OUT/bench.err
timeout 600 python tools/db_full_parity.py > $OUT/db_full_parity.json 2> $OUT/db_full_parity.err
timeout 120 python tools/host_io_rate.py > $OUT/host_io.jsonl 2>> $OUT/bench.err
timeout 200 python tools/stereo_rate.py > $OUT/stereo_rate.json 2>> $OUT/bench.err
timeout 200 python tools/matcher_latency.py > $OUT/matcher_latency.json 2>> $OUT/bench.err
timeout 300 python tools/bow_rate.py > $OUT/bow_rate.json 2>> $OUT/bench.err
timeout 300 python tools/camera_rate.py > $OUT/camera_rate.json 2>> $OUT/bench.err
timeout 60 ./tools/ta_ubench > $OUT/ta_ubench.txt 2>&1
timeout 60 ./tools/mfma_probe > $OUT/mfma_probe.txt 2>&1; timeout 60 ./tools/lds_dma_probe >> $OUT/mfma_probe.txt 2>&1
bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1
# the N > 1 code path on this 1-GPU box: two ranks share GPU 0 (B = 256 each = the default 512 frames per step in all), the launcher's rendezvous on 127.0.0.1
ORB_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 50 --warmup 3 --repeats 3 --batch 256 --no-cpu-baseline --no-host-io > $OUT/bench_2rank_shared.json 2>> $OUT/bench.err
timeout 300 python bench.py --pool --steps 20 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/bench_pool.json 2>> $OUT/bench.err
bash tools/gpu_single.sh $TAG/single > $OUT/single.log 2>&1
timeout 600 python tests/test_fuzz_gpu.py 120 77 > $OUT/gpu_fuzz_120cases.txt 2>&1
timeout 120 python tools/db_query_rate.py > $OUT/db_query_rate.jsonl 2>> $OUT/bench.err
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; cut -c1-1500 $OUT/bench.json; head -12 $OUT/kernel_stats.csv
python3 - <<PY
import json
for l in open('$OUT/bench_sweep.jsonl'):
    d=json.loads(l); print(d['config']['frames_per_step_per_gpu'], d['value'], d['ms_per_step'])
for f in ('bench_serial_b256','bench_sched1','bench_sched2','bench_sched3','bench_blur_valu'):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], {k:v for k,v in d['kernels_ms_per_launch'].items() if v})
    except Exception as e: print(f,'failed',e)
PY
cat $OUT/db_full_parity.json | cut -c1-400; tail -3 $OUT/bench.err; ls $OUT/pmc | head -3; cut -c1-300 $OUT/bench_2rank_shared.json; cut -c1-300 $OUT/bench_pool.json; tail -3 $OUT/single.log; tail -2 $OUT/gpu_fuzz_120cases.txt; cat $OUT/db_query_rate.jsonl | cut -c1-200
