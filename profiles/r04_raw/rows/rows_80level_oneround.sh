for rep in 1 2; do
python profiles/rows_sweep.py --dtype f64 --ni 4096 --nk 80 --nj 2048 --rows 64,128,1024 --rounds 4 2>&1 | grep -v amdgpu.ids
python profiles/rows_sweep.py --dtype f32 --ni 8192 --nk 80 --nj 2048 --rows 64,128,1024 --rounds 4 2>&1 | grep -v amdgpu.ids
python profiles/rows_sweep.py --dtype f32 --ni 8192 --nk 80 --nj 3072 --rows 64,128,1536 --rounds 4 2>&1 | grep -v amdgpu.ids
done
