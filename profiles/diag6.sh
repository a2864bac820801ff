#!/bin/bash
# r03 pass 6 (GPU box): column-wave pricing builds, WRF-native rows, evidence collection
set -u
export TMPDIR=/tmp
O=gpurun_out/diag6; mkdir -p $O
D=wrf-model-cuda-sample_amd/csrc/build/diag
P=wrf-model-cuda-sample_amd/libamt_advance_mu_t.so
python3 profiles/ab_libs.py --dtype f32 --ni 8192 --nk 80 --nj 4096 $P $D/libamt_halfchain.so $D/libamt_nochain.so $D/libamt_fewbar.so $D/libamt_fewbar_halfchain.so > $O/price_f32_80.txt 2>&1
python3 profiles/ab_libs.py --dtype f64 --ni 4096 --nk 80 --nj 2048 $P $D/libamt_halfchain.so $D/libamt_nochain.so $D/libamt_fewbar.so $D/libamt_fewbar_halfchain.so > $O/price_f64_80.txt 2>&1
python3 profiles/ab_libs.py --dtype f64 --ni 4096 --nk 60 --nj 4096 $P $D/libamt_halfchain.so $D/libamt_nochain.so $D/libamt_fewbar.so $D/libamt_fewbar_halfchain.so > $O/price_f64_60.txt 2>&1
python3 profiles/ab_libs.py --dtype f64 --ni 4096 --nk 128 --nj 1536 $P $D/libamt_halfchain.so $D/libamt_nochain.so $D/libamt_fewbar.so > $O/price_f64_128.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 4096 --unaligned --inner 4 auto 0,0,0,-1,1,36 0,0,0,-1,1,32 > $O/unaligned_f64.txt 2>&1
python3 profiles/ab_shapes.py --dtype f32 --ni 4095 --nk 60 --nj 4096 --unaligned --inner 4 auto 0,0,0,-1,1,36 > $O/unaligned_f32.txt 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_13_fullsize.py tests/test_gpu_90_bench_multirank.py tests/test_gpu_10_parity.py tests/test_gpu_12_random.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
bash profiles/collect.sh r03_f64_4096x60x4096 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-box-probe --probe-placements 1 > $O/collect_f64.log 2>&1
bash profiles/collect_slab.sh > $O/collect_slab.log 2>&1
mkdir -p gpurun_out/profiles_r03; cp profiles/r03_*kernel_stats.csv profiles/r03_*pmc.json profiles/hbm_traffic.json gpurun_out/profiles_r03/ 2>/dev/null
rocprofv3 --output-format json --kernel-trace --pmc TCC_EA0_RDREQ -d $O/chan_json -o chan -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-box-probe --probe-placements 1 > $O/chan_json.log 2>&1
ls -la $O/chan_json/*/ 2>/dev/null | head; 
tail -6 $O/price_*.txt $O/unaligned_*.txt | cut -c1-160; tail -3 $O/pytest.log; tail -5 $O/collect_slab.log
