#!/bin/bash
# r03 (GPU box): SQ / TA / TCP / TCC counters of the 60-level kernel against the level-group kernels (what do the 80-level
# shapes lose?).  One rocprofv3 --pmc pass per counter group and workload (profiles/pmc.sh); output: gpurun_out/counters/*.txt
set -u
O=gpurun_out/counters; mkdir -p $O
B="--steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-box-probe --probe-placements 1"
declare -A W
W[f64_60]="--ni 4096 --nk 60 --nj 4096"
W[f64_80]="--ni 4096 --nk 80 --nj 2048"
W[f32_80]="--dtype f32 --ni 8192 --nk 80 --nj 4096"
W[f32_60]="--dtype f32 --ni 4096 --nk 60 --nj 4096"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G3="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G4="TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
for w in f64_60 f64_80 f32_80 f32_60; do
  for g in 1 2 3 4; do
    eval "C=\$G$g"
    bash profiles/pmc.sh ${w}_g$g "$C" ${W[$w]} $B > $O/${w}_g$g.txt 2>&1
  done
done
tail -n 12 $O/*.txt
