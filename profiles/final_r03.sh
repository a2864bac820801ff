#!/bin/bash
# r03 final pass (GPU box): the driver's three steps on the final build + a kernel trace of the driver's bench command
set -u
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
( time timeout 3000 python3 -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_traced.json 2> $O/bench_traced.err
find $O/trace -name "*kernel_stats.csv" | head -2
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
tail -3 $O/pytest.log; cat $O/smoke.log | tail -2; tail -c 300 $O/bench_default.json; head -5 $O/kernel_stats.csv | cut -c1-200
