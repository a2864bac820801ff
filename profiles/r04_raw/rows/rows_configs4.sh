python profiles/rows_sweep.py --dtype f32 --ni 8192 --nk 80 --nj 8192 --rows 128,256,342,683 --rounds 3 --reps 5 2>&1 | grep -v amdgpu.ids
python profiles/rows_sweep.py --dtype f32 --ni 8192 --nk 80 --nj 4096 --rows 128,342,683 --rounds 3 --reps 5 2>&1 | grep -v amdgpu.ids
python profiles/rows_sweep.py --dtype f64 --ni 4096 --nk 60 --nj 8192 --rows 128,512,1024 --rounds 3 --reps 5 2>&1 | grep -v amdgpu.ids
