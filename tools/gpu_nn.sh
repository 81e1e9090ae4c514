#!/bin/bash
# brute-force descriptor DB (config 5): parity + rate, matrix-core scan vs popcount kernel.  usage: tools/gpu_nn.sh <tag>
TAG=${1:-nn}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_parity_match.py tests/test_full_size_gpu.py tests/test_host_pipeline.py -m gpu -q -x -k "brute or nn or db or pool or shard" 2>&1 | tail -2
timeout 600 python tools/db_full_parity.py > $OUT/db_full_parity_mfma.json 2> $OUT/err.log; cut -c1-700 $OUT/db_full_parity_mfma.json
timeout 300 python tools/db_query_rate.py > $OUT/db_query_mfma.json 2>> $OUT/err.log; cat $OUT/db_query_mfma.json
ORBHIP_NN=valu timeout 300 python tools/db_query_rate.py > $OUT/db_query_valu.json 2>> $OUT/err.log; cat $OUT/db_query_valu.json
tail -3 $OUT/err.log
