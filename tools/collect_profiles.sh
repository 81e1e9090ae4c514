#!/bin/bash
# copy the summaries of one tools/gpu_final.sh call (gpurun_out/<tag>) into profiles/ under a round prefix.   usage: tools/collect_profiles.sh <tag> <prefix>
TAG=$1; R=$2
cd "$(dirname "$0")/.."
S=gpurun_out/$TAG; D=profiles
cp $S/bench.json $D/${R}_bench.json
cp $S/kernel_stats.csv $D/${R}_rocprofv3_kernel_stats.csv
cp $S/bench_sweep.jsonl $D/${R}_bench_batch_sweep.jsonl
cp $S/bench_serial_b256.json $D/${R}_bench_serial_b256_standalone_kernels.json
cat $S/bench_sched1.json $S/bench_sched2.json $S/bench_sched3.json > $D/${R}_bench_schedules_1_2_3.jsonl
cp $S/bench_blur_valu.json $D/${R}_bench_blur_valu.json
cp $S/db_full_parity.json $D/${R}_db_full_parity_config5.json
cp $S/db_query_rate.jsonl $D/${R}_db_query_config5.jsonl
cp $S/host_io.jsonl $D/${R}_host_io.jsonl
cp $S/stereo_rate.json $D/${R}_stereo_rate.json
cp $S/bow_rate.json $D/${R}_bow_rate.json
cp $S/camera_rate.json $D/${R}_camera_rate.json
cp $S/matcher_latency.json $D/${R}_matcher_call_latency.json
cp $S/ta_ubench.txt $D/${R}_ta_cost_by_access_shape_ubench.txt
cp $S/mfma_probe.txt $D/${R}_mfma_and_lds_dma_probes.txt
cp $S/bench_2rank_shared.json $D/${R}_bench_2rank_shared_gpu.json
cp $S/bench_pool.json $D/${R}_bench_pool_one_process.json
cp $S/gpu_fuzz_120cases.txt $D/${R}_gpu_fuzz_120cases.txt
cp $S/box.txt $D/${R}_box.txt
( tail -4 $S/pytest_gpu.log; tail -2 $S/smoke.log ) > $D/${R}_pytest_gpu_tail.txt
( cat $S/single/plain.txt; cat $S/single/s_kernel_stats.csv; cat $S/single/s_memory_copy_stats.csv ) > $D/${R}_single_frame_call_timeline.txt 2>/dev/null
python3 tools/pmc_summarize.py $S/pmc $R
ls $D | grep "^${R}_" | wc -l
