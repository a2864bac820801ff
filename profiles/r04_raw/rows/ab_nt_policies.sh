D=wrf-model-cuda-sample_amd/csrc/build/diag
for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 8192 --nk 80 --nj 2048" "--dtype f64 --ni 4096 --nk 80 --nj 2048"; do
 echo "== $cfg"; python profiles/ab_libs.py $cfg --rounds 4 $D/libamt_base.so $D/libamt_ntst.so $D/libamt_ntld0.so $D/libamt_ntld2.so $D/libamt_ntdma.so 2>&1 | grep -v amdgpu.ids
done
