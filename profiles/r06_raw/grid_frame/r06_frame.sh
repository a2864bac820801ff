set -u
export TMPDIR=/tmp
O=gpurun_out/r06_frame; mkdir -p $O
D=wrf-model-cuda-sample_amd/csrc/build/diag
python3 profiles/ab_libs.py $D/libamt_head_c50e8bd.so wrf-model-cuda-sample_amd/libamt_advance_mu_t.so > $O/ab_headline.txt 2>&1
python3 profiles/ab_libs.py --dtype f32 --ni 8192 --nk 80 --nj 2048 $D/libamt_head_c50e8bd.so wrf-model-cuda-sample_amd/libamt_advance_mu_t.so > $O/ab_f32.txt 2>&1
python3 profiles/ab_libs.py --nk 80 --nj 2048 $D/libamt_head_c50e8bd.so wrf-model-cuda-sample_amd/libamt_advance_mu_t.so > $O/ab_f64_80.txt 2>&1
( timeout 900 python3 -m pytest tests/test_gpu_33_grid_native.py tests/test_gpu_34_halo_freshness.py tests/test_gpu_31_grid.py tests/test_gpu_30_slab_native.py -x -q -m gpu ) > $O/pytest_frame.log 2>&1; echo "rc $?" >> $O/pytest_frame.log
( AMT_GRID_FRAME=0 timeout 900 python3 -m pytest tests/test_gpu_33_grid_native.py -x -q -m gpu ) > $O/pytest_noframe.log 2>&1; echo "rc $?" >> $O/pytest_noframe.log
for N in 2048 1024 512; do
  python3 profiles/grid_loopback.py --ni $N --nj $N > $O/loop_frame_$N.txt 2>&1
  AMT_GRID_FRAME=0 python3 profiles/grid_loopback.py --ni $N --nj $N > $O/loop_noframe_$N.txt 2>&1
done
tail -n 12 $O/ab_headline.txt $O/ab_f32.txt $O/ab_f64_80.txt; tail -n 4 $O/pytest_frame.log $O/pytest_noframe.log; grep -h "patch\|stepper" $O/loop_*.txt
