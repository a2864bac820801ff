#!/bin/bash
# profiles/collect.sh TAG [bench args...] -- run ON THE GPU BOX (through gpurun) from the repo root.
# Writes raw rocprofv3 output under gpurun_out/prof_TAG/ and the summaries that get committed
# (profiles/summarize.py turns them into profiles/TAG_*.{csv,json} and refreshes profiles/hbm_traffic.json,
# the table bench.py reports as roofline.traffic).
#   pass 1: --kernel-trace --stats                 (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE                       (own pass: TCC has 4 slots, FETCH_SIZE takes 3)
#   pass 3: --pmc WRITE_SIZE
#   pass 4: --pmc TCC_HIT_sum TCC_MISS_sum         (L2 hit rate)
#   pass 5/6: the same two byte counters on amt_calib_stream_copy (known byte count, calibration)
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 5 --warmup 1 --no-cpu-baseline --no-verify}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export AMT_MARCH_VERBOSE=1     # the launcher's plan (kernel, waves, LDS, grid) goes into the logs
R="rocprofv3 --output-format csv"
$R --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
$R --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.log 2>&1
$R --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python3 bench.py $ARGS > $OUT/bench_write.log 2>&1
$R --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 -o l2 -- python3 bench.py $ARGS > $OUT/bench_l2.log 2>&1
$R --kernel-trace --pmc FETCH_SIZE -d $OUT/calib_fetch -o calib -- python3 profiles/calib.py > $OUT/calib_fetch.log 2>&1
$R --kernel-trace --pmc WRITE_SIZE -d $OUT/calib_write -o calib -- python3 profiles/calib.py > $OUT/calib_write.log 2>&1
python3 bench.py $ARGS > $OUT/bench_plain.log 2>&1
find $OUT -name "*.csv" | head -50
python3 profiles/summarize.py $TAG $OUT
