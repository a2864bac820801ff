# new launcher choice (rows 0) against 64-row blocks, three processes (= three placements) per size
for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 4096 --nk 60 --nj 4096" "--dtype f64 --ni 4096 --nk 80 --nj 2048" "--dtype f32 --ni 8192 --nk 80 --nj 2048" "--dtype f64 --ni 4096 --nk 60 --nj 512" "--dtype f64 --ni 2048 --nk 60 --nj 2048" "--dtype f64 --ni 1024 --nk 60 --nj 1024" "--dtype f64 --ni 512 --nk 60 --nj 512" "--dtype f64 --ni 1000 --nk 50 --nj 3000" "--dtype f32 --ni 1500 --nk 45 --nj 1200"; do
 for rep in 1 2 3; do python profiles/rows_sweep.py $cfg --rows 0,64 --rounds 4 2>&1 | grep -v amdgpu.ids; done
done
