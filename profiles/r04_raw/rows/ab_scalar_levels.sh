# per-level constants (dnw, fnm, fnp, rdnw) of the HL = 1 shapes through the scalar cache instead of the LDS table
D=wrf-model-cuda-sample_amd/csrc/build/diag
for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 4096 --nk 60 --nj 4096" "--dtype f64 --ni 4096 --nk 60 --nj 512" "--dtype f64 --ni 512 --nk 60 --nj 512"; do
 echo "== $cfg"; python profiles/ab_libs.py $cfg --rounds 5 $D/libamt_base.so $D/libamt_scal.so 2>&1 | grep -v amdgpu.ids
done
