#!/bin/bash
# r03 pass 7 (GPU box): residency cache (tests + timing), WRF-native rows with window-anchored tiles, full-size tests
set -u
O=gpurun_out/diag7; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_20_host_cache.py tests/test_gpu_13_fullsize.py tests/test_gpu_10_parity.py tests/test_gpu_12_random.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 profiles/oneshot.py --ni 1024 --nk 60 --nj 1024 > $O/oneshot_1024_f64.json 2> $O/oneshot_1024_f64.err
python3 profiles/oneshot.py --ni 512 --nk 60 --nj 512 > $O/oneshot_512_f64.json 2> $O/oneshot_512_f64.err
python3 profiles/oneshot.py --ni 2048 --nk 60 --nj 2048 > $O/oneshot_2048_f64.json 2> $O/oneshot_2048_f64.err
python3 profiles/oneshot.py --ni 1024 --nk 80 --nj 1024 --dtype f32 > $O/oneshot_1024_f32.json 2> $O/oneshot_1024_f32.err
python3 profiles/oneshot.py --ni 128 --nk 60 --nj 128 --reps 10 > $O/oneshot_128_f64.json 2> $O/oneshot_128_f64.err
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 4096 --unaligned --inner 4 auto 0,0,0,-1,1,36 0,0,0,-1,1,32 > $O/unaligned_f64.txt 2>&1
python3 profiles/ab_shapes.py --dtype f32 --ni 4095 --nk 60 --nj 4096 --unaligned --inner 4 auto 0,0,0,-1,1,36 > $O/unaligned_f32.txt 2>&1
python3 profiles/ab_shapes.py --ni 4096 --nk 60 --nj 4096 --inner 4 auto > $O/aligned_f64.txt 2>&1
tail -3 $O/pytest.log; cat $O/oneshot_*.json | cut -c1-900; tail -n 4 $O/unaligned_*.txt $O/aligned_f64.txt | cut -c1-170
