#!/bin/bash
# profiles/r05_slab_order.sh -- GPU box: one rank of 8 (4096x60x512 fp64) in loopback: what decides whether the exchange
# hides behind the interior.  Matrix: transport / pull mode x exchange enqueued before or after the interior x least rounds of
# the interior launch; each with the neighbour skew sweep.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_slab_order; mkdir -p $O
SK="0 200 500 1000"
run() { # name env...
  local name=$1; shift
  env "$@" python3 profiles/slab_loopback.py --nj 512 --sweeps 100 --skew-us $SK ${EXTRA:-} > $O/$name.txt 2>&1
}
for R in 2 4 8 16; do
  EXTRA="--transport ipc --beside-rounds $R" run ipc_fused_first_r$R AMT_SLAB_SKEW_WGS=1
done
EXTRA="--transport ipc --beside-rounds 2" run ipc_fused_after_r2 AMT_SLAB_EXCHANGE_FIRST=0
EXTRA="--transport ipc --beside-rounds 8" run ipc_fused_after_r8 AMT_SLAB_EXCHANGE_FIRST=0
EXTRA="--transport ipc --beside-rounds 2" run ipc_engine_first_r2 AMT_IPC_PULL=engine
EXTRA="--transport ipc --beside-rounds 8" run ipc_engine_first_r8 AMT_IPC_PULL=engine
EXTRA="--transport ipc --beside-rounds 8" run ipc_fused1wg_first_r8 AMT_IPC_PULL_WGS=1
EXTRA="--transport rccl --beside-rounds 2" run rccl_after_r2 AMT_SLAB_SKEW_WGS=31
EXTRA="--transport rccl --beside-rounds 8" run rccl_after_r8 AMT_SLAB_SKEW_WGS=31
EXTRA="--transport rccl --beside-rounds 2" run rccl_first_r2 AMT_SLAB_SKEW_WGS=31 AMT_SLAB_EXCHANGE_FIRST=1
EXTRA="--transport rccl --beside-rounds 8" run rccl_first_r8 AMT_SLAB_SKEW_WGS=31 AMT_SLAB_EXCHANGE_FIRST=1
for f in $O/*.txt; do echo "== $f"; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $f; done
