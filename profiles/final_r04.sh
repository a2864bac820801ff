#!/bin/bash
# r04 final pass (GPU box): a COLD box (no -march=native CPU libraries), the driver's three steps on the final build, a kernel
# trace of the driver's bench command, the configs[4] bench line, and the rocprofv3 summaries of the four reference workloads.
set -u
export TMPDIR=/tmp
O=gpurun_out/final4; mkdir -p $O
rm -rf oracle/_native
( time timeout 3000 python3 -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
( time python3 bench.py ) > $O/bench_noargs.json 2> $O/bench_noargs.err
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_traced.json 2> $O/bench_traced.err
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
python3 bench.py --dtype f32 --ni 8192 --nk 80 --nj 8192 --steps 10 --warmup 3 --probe-placements 1 --cpu-seconds 20 > $O/bench_configs4.json 2> $O/bench_configs4.err
tail -3 $O/pytest.log; tail -2 $O/smoke.log; tail -c 400 $O/bench_default.json; head -4 $O/kernel_stats.csv | cut -c1-200; tail -c 300 $O/bench_configs4.json
bash profiles/collect.sh r04_f64_4096x60x4096 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe --probe-placements 1 > $O/collect_a.log 2>&1
bash profiles/collect.sh r04_f32_8192x80x8192 --dtype f32 --ni 8192 --nk 80 --nj 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe --probe-placements 1 > $O/collect_b.log 2>&1
bash profiles/collect.sh r04_f64_4096x80x2048 --nk 80 --nj 2048 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe --probe-placements 1 > $O/collect_c.log 2>&1
bash profiles/collect.sh r04_f32_4096x60x4096 --dtype f32 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe --probe-placements 1 > $O/collect_d.log 2>&1
ls profiles/r04_* | head -20
cp profiles/r04_*kernel_stats.csv profiles/r04_*pmc.json profiles/hbm_traffic.json $O/ 2>/dev/null
