#!/bin/bash
# profiles/trace.sh TAG [bench args...] -- kernel-trace pass only (per-kernel durations), for the
# patch-sized launches where only the kernel time matters; run ON THE GPU BOX from the repo root.
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export AMT_MARCH_VERBOSE=1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py "$@" --no-cpu-baseline > $OUT/bench_trace.log 2>&1
python3 bench.py "$@" --no-cpu-baseline > $OUT/bench_plain.log 2>&1
python3 profiles/summarize.py $TAG $OUT | head -8
