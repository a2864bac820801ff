# start the workgroups a fraction of a row apart (lid % 16 steps of 0.9 us / 4 us; a row takes 14.7 us) instead of all at once
D=wrf-model-cuda-sample_amd/csrc/build/diag
for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 4096 --nk 60 --nj 4096" "--dtype f64 --ni 4096 --nk 60 --nj 512"; do
 echo "== $cfg"; python profiles/ab_libs.py $cfg --rounds 4 $D/libamt_base.so $D/libamt_stag90.so $D/libamt_stag400.so 2>&1 | grep -v amdgpu.ids
done
