# RCCL takes 31 workgroups (channels) for the halo send/recv group; a march workgroup fills a CU, so each of them needs a CU of its own.
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
for ch in default 16 8 4 2 1; do
  echo "== NCCL_MAX_NCHANNELS=$ch"
  if [ $ch = default ]; then python profiles/slab_loopback.py --nj 512 --sweeps 40 --skew-us 0 200 2>&1 | grep -v "$F"
  else NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1 python profiles/slab_loopback.py --nj 512 --sweeps 40 --skew-us 0 200 2>&1 | grep -v "$F"; fi
done
