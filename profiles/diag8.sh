#!/bin/bash
# r03 pass 8 (GPU box): the whole GPU suite as the driver runs it, then the evidence passes of the other workloads
set -u
export TMPDIR=/tmp
O=gpurun_out/diag8; mkdir -p $O
( time timeout 3000 python3 -m pytest tests -x -q -m gpu -s ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
bash profiles/collect.sh r03_f32_8192x80x8192 --dtype f32 --ni 8192 --nk 80 --nj 8192 --steps 4 --warmup 1 --no-cpu-baseline --no-verify --no-box-probe --probe-placements 1 > $O/collect_f32_80.log 2>&1
bash profiles/collect.sh r03_f64_4096x80x2048 --nk 80 --nj 2048 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-box-probe --probe-placements 1 > $O/collect_f64_80.log 2>&1
bash profiles/collect.sh r03_f32_4096x60x4096 --dtype f32 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-box-probe --probe-placements 1 > $O/collect_f32_60.log 2>&1
mkdir -p gpurun_out/profiles_r03; cp profiles/r03_*kernel_stats.csv profiles/r03_*pmc.json profiles/hbm_traffic.json gpurun_out/profiles_r03/ 2>/dev/null
python3 bench.py --dtype f32 --ni 8192 --nk 80 --nj 8192 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_configs4.json 2> $O/bench_configs4.err
grep -n "passed\|failed\|rc \|real\|random campaign\|streamed one-shot\|one-shot call" $O/pytest.log | head -20; tail -c 1500 $O/bench_configs4.json
