#!/bin/bash
# tests/test_concurrency.py (three caller threads in the binding: Tracking's loop, LocalMapping's and LoopClosing's matcher calls) with the reference's
# callers, the drop-in classes, the emitted matcher bodies and the CPU emulation of the kernels ALL compiled under AddressSanitizer, then under
# ThreadSanitizer.  usage: tools/sanitize_concurrency.sh [iterations per thread, default 200]      log: profiles/r05_sanitize_concurrency.txt (SANITIZE_LOG overrides the name)
cd "$(dirname "$0")/.."
ITERS=${1:-200}
LOG=profiles/${SANITIZE_LOG:-r05_sanitize_concurrency.txt}
make -C orb_slam2_amd/csrc -s emu_asan emu_tsan 2>&1 | grep -E "error" | head -3
make -C oracle -s dropin_asan dropin_tsan 2>&1 | grep -E "error" | head -3
{ echo "# tests/test_concurrency.py::test_three_threads_on_the_emulation, $ITERS calls on each of threads L and C beside T's loop rounds"; date -u; } > $LOG
run() {   # name, sanitizer runtime, library, options variable, options
    echo "## $1: LD_PRELOAD=$2 ORBSLAM_DROPIN_FULL_LIB=$3 $4=$5" >> $LOG
    env LD_PRELOAD=$(gcc -print-file-name=$2) ORBSLAM_DROPIN_FULL_LIB=$(pwd)/$3 $4="$5" ORBHIP_CONCURRENCY_ITERS=$ITERS ORBHIP_NO_MAKE=1 \
        timeout 3000 python -m pytest tests/test_concurrency.py -q -m "not gpu" -p no:cacheprovider > /tmp/sanitize_$1.log 2>&1
    echo "exit $?; reports: $(grep -c -E 'ERROR: AddressSanitizer|WARNING: ThreadSanitizer' /tmp/sanitize_$1.log)" >> $LOG
    grep -E "WARNING: ThreadSanitizer|ERROR: AddressSanitizer|SUMMARY" /tmp/sanitize_$1.log | sort | uniq -c | sort -rn | head -20 >> $LOG
    tail -3 /tmp/sanitize_$1.log | cut -c1-300 >> $LOG
}
# alloc_dealloc_mismatch=0: the reference-compiled checker libraries mix two cv::Mat stand-ins (new[] in one, free in the other) - test infrastructure, not product code
run asan libasan.so oracle/_ref/liborbslam_dropin_full_asan.so ASAN_OPTIONS detect_leaks=0:abort_on_error=0:detect_stack_use_after_return=0:alloc_dealloc_mismatch=0
run tsan libtsan.so oracle/_ref/liborbslam_dropin_full_tsan.so TSAN_OPTIONS "halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$(pwd)/tools/tsan.supp"
cat $LOG
