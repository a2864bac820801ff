# the neighbours' rows late AND the waiting exchange holding 31 CUs (AMT_SLAB_SKEW_WGS=31: RCCL's footprint) -- rows per interior block
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
for nj in 512 2048; do
for rows in 0 64 32 16 8; do
    echo "== nj $nj interior rows $rows (0 = launcher), 31 CUs held"
    AMT_SLAB_SKEW_WGS=31 AMT_MARCH_JROWS=$rows python profiles/slab_loopback.py --nj $nj --sweeps 40 --skew-us 0 100 200 300 500 2>&1 | grep -v "$F" | grep -v "native stepper"
done
done
