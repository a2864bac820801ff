#!/bin/bash
# r05 final pass (GPU box): a COLD box (no -march=native CPU libraries), the driver's three steps on the final build, a kernel
# trace of the driver's bench command, WRF's unpadded rows, and the rocprofv3 summaries of the reference workloads.
set -u
export TMPDIR=/tmp
O=gpurun_out/final5; mkdir -p $O
rm -rf oracle/_native
( time timeout 3000 python3 -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
( time python3 bench.py ) > $O/bench_noargs.json 2> $O/bench_noargs.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_traced.json 2> $O/bench_traced.err
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done; rm -rf $O/trace
python3 bench.py --align-elems 1 --no-cpu-baseline > $O/bench_rows4098.json 2> $O/bench_rows4098.err
python3 bench.py --gpus 2 --share-gpu --transport ipc --steps 5 --warmup 2 --no-box-probe > $O/bench_share2_ipc.json 2> $O/bench_share2_ipc.err
tail -3 $O/pytest.log; tail -2 $O/smoke.log; tail -c 300 $O/bench_noargs.json; head -4 $O/kernel_stats.csv | cut -c1-200
bash profiles/collect.sh r05_f64_4096x60x4096 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe > $O/collect_a.log 2>&1
bash profiles/collect.sh r05_f64_4096x60x4096_rows4098 --align-elems 1 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe > $O/collect_b.log 2>&1
cp profiles/r05_*kernel_stats.csv profiles/r05_*pmc.json profiles/hbm_traffic.json $O/ 2>/dev/null
ls $O
