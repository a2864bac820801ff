D=wrf-model-cuda-sample_amd/csrc/build/diag; L=wrf-model-cuda-sample_amd/libamt_advance_mu_t.so
run() { echo "== $1"; python profiles/ab_libs.py $1 --rounds 5 $L $2 $3 $4 2>&1 | grep -v amdgpu.ids; }
run "--dtype f64 --ni 4096 --nk 60 --nj 1024" $D/libamt_j128.so:AMT_MARCH_JROWS=128 $D/libamt_j256.so:AMT_MARCH_JROWS=256
run "--dtype f64 --ni 4096 --nk 60 --nj 2048" $D/libamt_j128.so:AMT_MARCH_JROWS=128 $D/libamt_j256.so:AMT_MARCH_JROWS=256 $D/libamt_j512.so:AMT_MARCH_JROWS=512
run "--dtype f64 --ni 2048 --nk 60 --nj 2048" $D/libamt_j128.so:AMT_MARCH_JROWS=128 $D/libamt_j256.so:AMT_MARCH_JROWS=256
run "--dtype f32 --ni 4096 --nk 60 --nj 4096" $D/libamt_j256.so:AMT_MARCH_JROWS=256 $D/libamt_j512.so:AMT_MARCH_JROWS=512 $D/libamt_j1024.so:AMT_MARCH_JROWS=1024
run "--dtype f64 --ni 4096 --nk 60 --nj 4096" $D/libamt_j256.so:AMT_MARCH_JROWS=256 $D/libamt_j512.so:AMT_MARCH_JROWS=512 $D/libamt_j1024.so:AMT_MARCH_JROWS=1024
run "--dtype f64 --ni 4096 --nk 80 --nj 2048" $D/libamt_j128.so:AMT_MARCH_JROWS=128 $D/libamt_j256.so:AMT_MARCH_JROWS=256 $D/libamt_j512.so:AMT_MARCH_JROWS=512
run "--dtype f64 --ni 1024 --nk 60 --nj 1024" $D/libamt_j128.so:AMT_MARCH_JROWS=128 $D/libamt_j256.so:AMT_MARCH_JROWS=32
