for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096 --rows 64,256,1024" "--dtype f32 --ni 4096 --nk 60 --nj 4096 --rows 64,512" "--dtype f64 --ni 4096 --nk 80 --nj 2048 --rows 64,256" "--dtype f32 --ni 8192 --nk 80 --nj 2048 --rows 64,512" "--dtype f64 --ni 4096 --nk 60 --nj 512 --rows 64,128" "--dtype f64 --ni 2048 --nk 60 --nj 2048 --rows 64,256" "--dtype f64 --ni 1000 --nk 50 --nj 3000 --rows 0,188"; do
 python profiles/rows_sweep.py $cfg --xchunk 0,32,16,64 2>&1 | grep -v amdgpu.ids
done
