D=wrf-model-cuda-sample_amd/csrc/build/diag; L=wrf-model-cuda-sample_amd/libamt_advance_mu_t.so
for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 8192 --nk 80 --nj 2048" "--dtype f64 --ni 4096 --nk 80 --nj 2048"; do
 echo "== $cfg"; python profiles/ab_libs.py $cfg --rounds 5 $L $D/libamt_j96.so:AMT_MARCH_JROWS=96 $D/libamt_j128.so:AMT_MARCH_JROWS=128 $D/libamt_j256.so:AMT_MARCH_JROWS=256
done
