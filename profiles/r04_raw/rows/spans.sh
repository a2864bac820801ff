export AMT_LIBRARY=wrf-model-cuda-sample_amd/csrc/build/diag/libamt_spans.so
for a in "--rows 64 --xchunk 0" "--rows 64 --xchunk 32" "--rows 1024 --xchunk 0" "--rows 512 --xchunk 0"; do python profiles/spans.py --dtype f64 $a 2>&1 | grep -v amdgpu.ids; done
for a in "--rows 64 --xchunk 0" "--rows 64 --xchunk 32" "--rows 256 --xchunk 0"; do python profiles/spans.py --dtype f64 --nk 80 --nj 2048 $a 2>&1 | grep -v amdgpu.ids; done
