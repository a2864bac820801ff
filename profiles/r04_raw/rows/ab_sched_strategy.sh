# hipcc's other machine-scheduler strategies (-mllvm -amdgpu-sched-strategy=...) on the march kernel
D=wrf-model-cuda-sample_amd/csrc/build/diag
for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 8192 --nk 80 --nj 2048"; do
 echo "== $cfg"; python profiles/ab_libs.py $cfg --rounds 5 $D/libamt_base.so $D/libamt_ilp.so $D/libamt_clause.so 2>&1 | grep -v amdgpu.ids
done
