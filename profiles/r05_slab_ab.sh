#!/bin/bash
# profiles/r05_slab_ab.sh -- GPU box: the j-slab sweep of ONE rank of 8 (4096x60x512 fp64) through the native stepper in
# loopback, RCCL against the IPC transport (VERDICT r04 item 1 iii).  The neighbour's lateness is emulated by the skew hook
# in front of the exchange; it is given the footprint of what WAITS in each transport: RCCL's send/recv kernel holds 31
# workgroups (AMT_SLAB_SKEW_WGS=31), the IPC transport's wait is one wave (AMT_SLAB_SKEW_WGS=1).  Then a kernel trace of
# each, for the per-kernel durations beside the interior.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_slab_ab; mkdir -p $O
SK="0 100 200 300 500 1000"
AMT_SLAB_SKEW_WGS=31 python3 profiles/slab_loopback.py --nj 512 --transport rccl --skew-us $SK > $O/rccl_wgs31.txt 2>&1
AMT_SLAB_SKEW_WGS=1  python3 profiles/slab_loopback.py --nj 512 --transport ipc  --skew-us $SK > $O/ipc_engine_wgs1.txt 2>&1
AMT_SLAB_SKEW_WGS=1 AMT_IPC_PULL=kernel python3 profiles/slab_loopback.py --nj 512 --transport ipc --skew-us $SK > $O/ipc_kernel_wgs1.txt 2>&1
for T in rccl ipc; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$T -o t -- python3 profiles/slab_loopback.py --nj 512 --transport $T --sweeps 50 > $O/trace_$T.log 2>&1
  f=$(find $O/trace_$T -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats.csv
done
AMT_IPC_PULL=kernel rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ipck -o t -- python3 profiles/slab_loopback.py --nj 512 --transport ipc --sweeps 50 > $O/trace_ipck.log 2>&1
f=$(find $O/trace_ipck -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/ipc_pullkernel_kernel_stats.csv
rm -rf $O/trace_rccl $O/trace_ipc $O/trace_ipck
tail -n 12 $O/rccl_wgs31.txt $O/ipc_engine_wgs1.txt $O/ipc_kernel_wgs1.txt
