#!/bin/bash
# profiles/r05_slab_reserve.sh -- GPU box: second matrix of the loopback slab (4096x60x512 fp64, one rank of 8): the interior
# planned to leave compute units free for the communication stream (amt_march_set_beside rounds, reserve).
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_slab_reserve; mkdir -p $O
SK="0 200 500 1000"
run() { local name=$1; shift; env "$@" python3 profiles/slab_loopback.py --nj 512 --sweeps 100 --skew-us $SK ${EXTRA:-} > $O/$name.txt 2>&1; }
for RR in "1 8" "2 8" "4 8" "8 8" "4 4" "4 16" "2 16"; do
  set -- $RR
  EXTRA="--transport ipc --beside-rounds $1 --beside-reserve $2" run ipc_fused4_r$1_res$2 AMT_IPC_PULL_WGS=4
done
EXTRA="--transport ipc --beside-rounds 4 --beside-reserve 16" run ipc_fused12_r4_res16 AMT_IPC_PULL_WGS=12
EXTRA="--transport ipc --beside-rounds 4 --beside-reserve 8" run ipc_engine_r4_res8 AMT_IPC_PULL=engine
EXTRA="--transport ipc --beside-rounds 8 --beside-reserve 8" run ipc_engine_r8_res8 AMT_IPC_PULL=engine
EXTRA="--transport rccl --beside-rounds 4 --beside-reserve 32" run rccl_after_r4_res32 AMT_SLAB_SKEW_WGS=31
EXTRA="--transport rccl --beside-rounds 4 --beside-reserve 8" run rccl4ch_after_r4_res8 AMT_SLAB_SKEW_WGS=4 NCCL_MAX_NCHANNELS=4 NCCL_MIN_NCHANNELS=1
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 profiles/slab_loopback.py --nj 512 --sweeps 10 --transport rccl > $O/rccl4ch_trace.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/tr/**/*kernel_trace.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "rccl" in r["Kernel_Name"] and (r["Grid_Size_X"], r["Workgroup_Size_X"]) not in seen:
            seen.add((r["Grid_Size_X"], r["Workgroup_Size_X"]))
            print("rccl kernel grid", r["Grid_Size_X"], "workgroup", r["Workgroup_Size_X"])
PY
rm -rf $O/tr
for f in $O/*.txt; do echo "== $f"; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $f; done
