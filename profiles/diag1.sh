#!/bin/bash
# r03 diagnostic pass 1 (GPU box): box vs placement; available per-channel counters
set -u
export TMPDIR=/tmp
O=gpurun_out/diag1; mkdir -p $O
rocm-smi --showclocks --showpower --showperflevel > $O/smi_start.txt 2>&1
( rocprofv3 --list-avail > $O/avail.txt 2>&1 || rocprofv3 -L > $O/avail.txt 2>&1 )
grep -n -i "TCC_EA0_RDREQ\|TCC_EA0_WRREQ\|TCC_REQ\b\|TCC_BUSY\|TCC_EA0_RD_UNCACHED\|MALL\|TCC_TAG_STALL\|TCC_EA0_RDREQ_DRAM\|TCC_EA0_WRREQ_DRAM" $O/avail.txt | head -80 > $O/avail_tcc.txt
for k in 1 2 3; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$k.json 2> $O/bench_$k.err
done
timeout 1500 python3 profiles/box_probe.py > $O/probe.jsonl 2> $O/probe.err
rocm-smi --showclocks --showpower --showperflevel > $O/smi_end.txt 2>&1
tail -c 600 $O/bench_1.json; echo; tail -2 $O/probe.jsonl | cut -c1-1500
