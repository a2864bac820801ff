#!/bin/bash
# r03 diagnostic pass 3 (GPU box): rows per workgroup with window-origin tiles
set -u
O=gpurun_out/diag3; mkdir -p $O
J() { echo "0,0,0,-1,1,$1"; }
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 512 --inner 20 auto $(J 8) $(J 11) $(J 16) $(J 19) $(J 22) $(J 26) $(J 32) $(J 43) $(J 64) > $O/j512.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 510 --inner 20 auto $(J 8) $(J 11) $(J 16) $(J 19) $(J 22) $(J 26) $(J 32) $(J 43) $(J 64) > $O/j510.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 1024 --inner 10 auto $(J 8) $(J 16) $(J 22) $(J 32) $(J 43) $(J 64) > $O/j1024.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 2048 --inner 6 auto $(J 16) $(J 32) $(J 43) $(J 64) > $O/j2048.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 4096 --inner 4 auto $(J 16) $(J 24) $(J 32) $(J 36) $(J 43) $(J 48) $(J 64) > $O/j4096.txt 2>&1
python3 profiles/ab_shapes.py --ni 512 --nk 60 --nj 512 --inner 50 auto $(J 4) $(J 8) $(J 11) $(J 16) $(J 22) $(J 32) 1,4,2,0,1,8 1,4,2,0,1,16 > $O/s512.txt 2>&1
python3 profiles/ab_shapes.py --ni 1024 --nk 60 --nj 1024 --inner 30 auto $(J 8) $(J 16) $(J 22) $(J 32) $(J 43) $(J 64) > $O/s1024.txt 2>&1
python3 profiles/ab_shapes.py --ni 256 --nk 60 --nj 256 --inner 100 auto $(J 1) $(J 2) $(J 4) $(J 8) 1,4,2,0,1,2 1,4,2,0,1,4 > $O/s256.txt 2>&1
python3 profiles/ab_shapes.py --dtype f32 --ni 8192 --nk 80 --nj 4096 --inner 3 auto $(J 16) $(J 32) $(J 36) $(J 64) > $O/f32_80.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 80 --nj 2048 --inner 4 auto $(J 16) $(J 32) $(J 36) $(J 64) > $O/f64_80.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 4096 --unaligned --inner 4 auto $(J 32) $(J 64) > $O/j4096_unaligned.txt 2>&1
tail -n 12 $O/*.txt
