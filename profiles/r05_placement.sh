#!/bin/bash
# profiles/r05_placement.sh -- GPU box: the placement experiment of profiles/r05_placement.py, plain and under three counter sets.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_placement; mkdir -p $O
python3 profiles/r05_placement.py --k 12 > $O/plain.txt 2>&1
i=0
for SET in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/pmc$i -o p -- python3 profiles/r05_placement.py --k 12 > $O/pmc$i.log 2>&1
  python3 profiles/r05_placement.py --report $O/pmc$i > $O/pmc$i.report.txt 2>&1
  rm -rf $O/pmc$i
done
cat $O/plain.txt | tail -14; for f in $O/pmc*.report.txt; do echo "== $f"; head -12 $f; done
