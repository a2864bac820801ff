# one rank of eight's slab (and of two), loopback: order of the exchange and the interior launch x rows per interior block,
# with the neighbours' rows (= the wire) 0 / 200 / 400 us late
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
for nj in 512 2048; do
for first in 0 1; do
  for rows in 0 64 32 16; do
    echo "== nj $nj exchange_first $first interior rows $rows (0 = launcher)"
    AMT_SLAB_EXCHANGE_FIRST=$first AMT_MARCH_JROWS=$rows python profiles/slab_loopback.py --nj $nj --sweeps 40 --skew-us 0 200 400 2>&1 | grep -v "$F"
  done
done
done
