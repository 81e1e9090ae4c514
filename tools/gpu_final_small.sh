#!/bin/bash
# kernel timeline of single-image calls + the quadtree's phase stamps on the in-tree library (build ab/liborbhip_qttrace.so first: tools/trace_builds.py qttrace)
cd "$(dirname "$0")/.."; R=$(pwd)
NROWS=13 tools/gpu_trace_variant.sh final tree | grep -v "^W2026"
ORBHIP_LIBRARY=$R/ab/liborbhip_qttrace.so python3 tools/qt_trace_experiment.py 2>&1 > gpurun_out/final/qt_phases.txt
cat gpurun_out/final/qt_phases.txt | cut -c1-200 | head -12
