#!/bin/bash
# profiles/r05_slab_hostwait.sh -- GPU box: the host-waited schedule of the IPC transport against the device-waited ones and RCCL,
# one rank of 8 (4096x60x512 fp64) in loopback, neighbour lateness 0 .. 1.5 ms, all on ONE box.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_slab_hostwait; mkdir -p $O
SK="0 200 500 1000 1500"
run() { local name=$1; shift; env "$@" python3 profiles/slab_loopback.py --nj 512 --sweeps 100 --skew-us $SK ${EXTRA:-} > $O/$name.txt 2>&1; }
EXTRA="--transport ipc" run ipc_hostwait_kernelpull AMT_IPC_HOST_WAIT=1
EXTRA="--transport ipc" run ipc_hostwait_engine AMT_IPC_HOST_WAIT=1 AMT_IPC_PULL=engine
EXTRA="--transport ipc" run ipc_devwait_r2 AMT_IPC_HOST_WAIT=0
EXTRA="--transport ipc --beside-rounds 4 --beside-reserve 16" run ipc_devwait_r4_res16 AMT_IPC_HOST_WAIT=0
EXTRA="--transport rccl" run rccl_r2_4cta AMT_SLAB_SKEW_WGS=4
EXTRA="--transport rccl" run rccl_r2_31cta AMT_SLAB_SKEW_WGS=31 AMT_RCCL_MAX_CTAS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 profiles/slab_loopback.py --nj 512 --sweeps 50 --transport ipc > $O/trace_hostwait.log 2>&1
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/ipc_hostwait_kernel_stats.csv; rm -rf $O/tr
for f in $O/*.txt; do echo "== $f"; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $f; done
head -8 $O/ipc_hostwait_kernel_stats.csv | cut -c1-200
