#!/bin/bash
# PMC passes over round 5's new kernels (tools/pmc_r05_workload.py), seven passes incl. the matrix-core counters
cd "$(dirname "$0")/.."
bash tools/gpu_pmc.sh r05_pmc_r05 7 r05 > gpurun_out/r05_pmc_r05.log 2>&1
tail -5 gpurun_out/r05_pmc_r05.log | cut -c1-600; ls gpurun_out/r05_pmc_r05 | head -30; tail -3 gpurun_out/r05_pmc_r05/p7.stderr
exit 0
