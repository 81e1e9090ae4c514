#!/bin/bash
# copy the summaries of one tools/gpu_final_short.sh call into profiles/ under a prefix.   usage: tools/collect_profiles_short.sh <tag> <prefix, e.g. r03b>
TAG=$1; R=$2
cd "$(dirname "$0")/.."
S=gpurun_out/$TAG; D=profiles
cp $S/bench.json $D/${R}_bench.json
cp $S/kernel_stats.csv $D/${R}_rocprofv3_kernel_stats.csv
cp $S/bench_sweep.jsonl $D/${R}_bench_batch_sweep.jsonl
cp $S/bench_2rank_shared.json $D/${R}_bench_2rank_shared_gpu.json
cp $S/gpu_fuzz_60cases.txt $D/${R}_gpu_fuzz_60cases.txt
cp $S/box.txt $D/${R}_box.txt
( tail -4 $S/pytest_gpu.log; tail -2 $S/smoke.log ) > $D/${R}_pytest_gpu_tail.txt
python3 tools/pmc_summarize.py $S/pmc $R
ls $D | grep "^${R}_" | wc -l
