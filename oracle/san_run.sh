#!/bin/sh
# TEST INFRASTRUCTURE: the CPU suite with the oracle built under AddressSanitizer + UndefinedBehaviorSanitizer
# (make -C oracle san).  The sanitizer runtimes are preloaded because the host program (python) is not instrumented.
set -e
cd "$(dirname "$0")/.."
make -C oracle san >/dev/null
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
AMT_ORACLE_LIBRARY="$PWD/oracle/_san/liboracle_amt.so" LD_PRELOAD="$ASAN $UBSAN" \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest ${@:-tests} -x -q -m "not gpu" -p no:cacheprovider
