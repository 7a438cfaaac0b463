#!/bin/bash
# run a python command against the sanitizer build of the library's own sources (tests/hostemu/build.py):
#   tests/hostemu/run.sh [gcc|clang|clang_pattern] python -m pytest tests/test_gpu_parity.py -m gpu -k lockstep -x -q
V=${1:-gcc}; shift
HERE=$(cd "$(dirname "$0")" && pwd)
case "$V" in
  gcc) PRE=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ;;
  *) PRE=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) ;;
esac
export SAG_LIB=$HERE/_build/libsag_hostemu_$V.so SAG_HOSTEMU=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:detect_stack_use_after_return=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=${HOSTEMU_HALT:-1}
LD_PRELOAD=$PRE exec "$@"
