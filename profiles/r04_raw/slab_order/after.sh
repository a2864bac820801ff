# after the change: the stepper's interior launch beside the exchange is planned in at least two rounds
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
for nj in 512 1024 2048; do
for first in 0 1; do
    echo "== nj $nj exchange_first $first"
    AMT_MARCH_VERBOSE=0 AMT_SLAB_EXCHANGE_FIRST=$first python profiles/slab_loopback.py --nj $nj --sweeps 40 --skew-us 0 100 200 400 800 2>&1 | grep -v "$F"
done
done
