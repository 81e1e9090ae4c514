#!/bin/bash
# The CPU suite with the kernel sources + host code of the library compiled under AddressSanitizer (fiber emulation build).
# usage: tools/asan_suite.sh [pytest args]     log: profiles/r05_asan_suite.txt (ASAN_LOG overrides the name)
cd "$(dirname "$0")/.."
make -C orb_slam2_amd/csrc -s emu_asan 2>&1 | grep -v -E "warning|note:|\^|~|\|" | tail -3
export ORBHIP_EMU_LIB=$(pwd)/tests/emu/asan/liborbhip_emu.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
# alloc_dealloc_mismatch=0: the reference-compiled checker libraries (oracle/_ref) mix two cv::Mat stand-ins (new[] in one, free in the other) — test infrastructure, not product code
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=0:alloc_dealloc_mismatch=0
LOG=profiles/${ASAN_LOG:-r05_asan_suite.txt}
{ echo "# python -m pytest tests -m 'not gpu' with ORBHIP_EMU_LIB=tests/emu/asan/liborbhip_emu.so (g++ -fsanitize=address) and libasan preloaded"; date -u; } > $LOG
python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@" > /tmp/asan_suite_full.log 2>&1; grep -c "ERROR: AddressSanitizer" /tmp/asan_suite_full.log | sed "s/^/AddressSanitizer reports: /" >> $LOG; tail -4 /tmp/asan_suite_full.log | cut -c1-300 >> $LOG
tail -6 $LOG
