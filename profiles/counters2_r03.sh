#!/bin/bash
# r03 (GPU box): SQ counters of the level-group kernels, march kernels only (--kernel-include-regex), one timed sweep
set -u
export TMPDIR=/tmp
O=gpurun_out/counters2; mkdir -p $O
B="--steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-box-probe --probe-placements 1"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {  # tag counters args...
  local tag=$1 cnt=$2; shift 2
  local t0=$(date +%s)
  timeout 600 rocprofv3 --output-format csv --kernel-trace --kernel-include-regex amt_march --pmc $cnt -d $O/$tag -o pmc -- python3 bench.py "$@" $B > $O/$tag.log 2>&1
  echo "$tag rc $? $(( $(date +%s) - t0 )) s" >> $O/times.txt
  python3 - "$O/$tag" <<'PY' > $O/$tag.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
}




G3="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G4="TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
G5="TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"
if [ "${1:-all}" = "sq" ]; then
run f64_80_g2 "$G2" --ni 4096 --nk 80 --nj 2048
run f64_80_g1 "$G1" --ni 4096 --nk 80 --nj 2048
run f64_60b_g2 "$G2" --ni 4096 --nk 60 --nj 2048
run f64_60b_g1 "$G1" --ni 4096 --nk 60 --nj 2048
run f64_84_g2 "$G2" --ni 4096 --nk 84 --nj 2048
fi
for g in 3 4 5; do
  eval "C=\$G$g"
  run f64_80_g$g "$C" --ni 4096 --nk 80 --nj 2048
  run f64_60b_g$g "$C" --ni 4096 --nk 60 --nj 2048
done
run f64_84_g2 "$G2" --ni 4096 --nk 84 --nj 2048
run f64_84_g1 "$G1" --ni 4096 --nk 84 --nj 2048
cat $O/times.txt; cat $O/*g3.txt $O/*g4.txt $O/*g5.txt $O/f64_84*.txt | head -120
