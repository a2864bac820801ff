#!/bin/bash
# r03 diagnostic pass 4 (GPU box): tapered block schedule -- parity, then A/B against uniform blocks
set -u
O=gpurun_out/diag4; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_10_parity.py tests/test_gpu_11_shapes.py tests/test_gpu_12_random.py tests/test_gpu_13_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
A="python3 profiles/ab_shapes.py"
$A --ni 4096 --nk 60 --nj 512 --inner 20 t0 t1 t48 t32 t24 t16 > $O/j512.txt 2>&1
$A --ni 4096 --nk 60 --nj 510 --inner 20 t0 t1 t32 > $O/j510.txt 2>&1
$A --ni 4096 --nk 60 --nj 1024 --inner 10 t0 t1 t48 t32 > $O/j1024.txt 2>&1
$A --ni 4096 --nk 60 --nj 2048 --inner 6 t0 t1 t48 t32 > $O/j2048.txt 2>&1
$A --ni 4096 --nk 60 --nj 4096 --inner 4 t0 t1 t96 t48 t32 > $O/j4096.txt 2>&1
$A --ni 2048 --nk 60 --nj 2048 --inner 10 t0 t1 t48 t32 t16 > $O/s2048.txt 2>&1
$A --ni 1024 --nk 60 --nj 1024 --inner 30 t0 t1 t32 t16 t8 > $O/s1024.txt 2>&1
$A --ni 512 --nk 60 --nj 512 --inner 50 t0 t1 t16 t8 > $O/s512.txt 2>&1
$A --dtype f32 --ni 8192 --nk 80 --nj 4096 --inner 3 t0 t1 t48 t32 > $O/f32_80.txt 2>&1
$A --ni 4096 --nk 80 --nj 2048 --inner 4 t0 t1 t48 t32 > $O/f64_80.txt 2>&1
$A --dtype f32 --ni 4096 --nk 60 --nj 4096 --inner 4 t0 t1 t48 t32 > $O/f32_60.txt 2>&1
$A --ni 4096 --nk 60 --nj 4096 --unaligned --inner 4 t0 t1 t32 > $O/j4096_unaligned.txt 2>&1
tail -3 $O/pytest.log; tail -n 8 $O/*.txt | cut -c1-150
