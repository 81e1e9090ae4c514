#!/bin/bash
# A/B measurements in ONE gpurun call (boxes differ by a few per cent): builds liborbhip.so of another git revision into ab/liborbhip_<name>.so.
# bench.py / the tools load it when ORBHIP_LIBRARY names it.     usage: tools/build_ref_lib.sh <git-ref> <name>
set -e
cd "$(dirname "$0")/.."
REF=${1:-HEAD}; NAME=${2:-base}
rm -rf /tmp/orbhip_ab && mkdir -p /tmp/orbhip_ab ab
git archive $REF orb_slam2_amd/csrc include | tar -x -C /tmp/orbhip_ab
make -C /tmp/orbhip_ab/orb_slam2_amd/csrc -s -j8 2>&1 | grep -E "error" || true
cp /tmp/orbhip_ab/orb_slam2_amd/liborbhip.so ab/liborbhip_$NAME.so
ls -la ab/
