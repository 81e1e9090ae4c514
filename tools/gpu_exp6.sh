#!/bin/bash
TAG=${1:-exp6}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_00_device.py tests/test_host_pipeline.py tests/test_parity_extract.py tests/test_dropin_cpp.py tests/test_reference_dropin.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
python3 tools/single_frame_calls.py > $OUT/single.txt 2>&1
timeout 600 python3 tools/host_io_matrix.py > $OUT/host_io_matrix.jsonl 2> $OUT/err.txt
tail -3 $OUT/pytest.log; cat $OUT/single.txt; cat $OUT/host_io_matrix.jsonl; tail -3 $OUT/err.txt
