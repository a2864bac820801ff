#!/bin/bash
# r04: A/B on the GPU box of (a) the LDS-DMA issued from inline assembly with explicit landing waits (pf0) against the
# builtin (r03 = the r03 kernel), and (b) the one-row-ahead register prefetch of P1's global loads on top of it
# (pf11: v_1, u, u_1).  A forced shape needs its own copy of the library (the AMT_MARCH_* variables are read once per
# loaded library).
O=gpurun_out/r4_pf
mkdir -p $O
D=wrf-model-cuda-sample_amd/csrc/build/diag
K4="AMT_MARCH_KPT=4,AMT_MARCH_WM=12"
AMT_LIBRARY=$PWD/$D/libamt_pf0.so timeout 900 python3 -m pytest tests/test_gpu_11_shapes.py tests/test_gpu_10_parity.py tests/test_gpu_12_random.py -x -q -m gpu > $O/parity_pf0.log 2>&1
echo "parity pf0: $(tail -1 $O/parity_pf0.log)"
ab() { python3 profiles/ab_libs.py "$@" 2>&1 | grep -v amdgpu.ids; }
ab --dtype f32 --ni 8192 --nk 80 --nj 2048 $D/libamt_r03.so $D/libamt_pf0.so $D/libamt_r03_k4w12.so:$K4 $D/libamt_pf0_k4w12.so:$K4 $D/libamt_pf11_k4w12.so:$K4 | tee $O/f32_80.txt
ab --dtype f64 --ni 4096 --nk 80 --nj 2048 $D/libamt_r03.so $D/libamt_pf0.so $D/libamt_r03_k4w12.so:$K4 $D/libamt_pf0_k4w12.so:$K4 | tee $O/f64_80.txt
ab --dtype f64 --ni 4096 --nk 60 --nj 4096 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f64_60.txt
ab --dtype f32 --ni 4096 --nk 60 --nj 4096 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f32_60.txt
ab --dtype f64 --ni 4096 --nk 40 --nj 4096 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f64_40.txt
ab --dtype f64 --ni 4096 --nk 30 --nj 4096 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f64_30.txt
ab --dtype f64 --ni 4096 --nk 128 --nj 1024 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f64_128.txt
ab --dtype f64 --ni 512 --nk 60 --nj 512 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f64_512.txt
ab --dtype f64 --ni 4096 --nk 60 --nj 512 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f64_slab.txt
ab --dtype f64 --ni 64 --nk 40 --nj 64 $D/libamt_r03.so $D/libamt_pf0.so | tee $O/f64_64.txt
