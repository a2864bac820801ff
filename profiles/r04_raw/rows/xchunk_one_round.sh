# one-round launches: which tiles an XCD gets (xchunk 0 = 32 neighbouring tiles of one j block; 16 = 16 tiles of two j blocks;
# 8 = 8 tiles of each of the four j blocks), three processes = three placements
for rep in 1 2 3; do
python profiles/rows_sweep.py --dtype f64 --ni 4096 --nk 60 --nj 4096 --rows 0 --xchunk 0,16,8,4 --rounds 4 2>&1 | grep -v amdgpu.ids
done
python profiles/rows_sweep.py --dtype f32 --ni 4096 --nk 60 --nj 4096 --rows 0 --xchunk 0,16,8 --rounds 4 2>&1 | grep -v amdgpu.ids
python profiles/rows_sweep.py --dtype f64 --ni 4096 --nk 60 --nj 512 --rows 0 --xchunk 0,16,8 --rounds 4 2>&1 | grep -v amdgpu.ids
