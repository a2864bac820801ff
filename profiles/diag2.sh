#!/bin/bash
# r03 diagnostic pass 2 (GPU box): window-origin tiles (parity + size sweep), placement by allocation API, streaming ceilings
set -u
export TMPDIR=/tmp
O=gpurun_out/diag2; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_10_parity.py tests/test_gpu_11_shapes.py tests/test_gpu_12_random.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 ./profiles/vmm_probe > $O/vmm.jsonl 2> $O/vmm.err
timeout 1200 python3 profiles/sweep.py > $O/size_sweep.md 2> $O/size_sweep.err
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ -d $O/chan_rd -o chan -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $O/chan_rd.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL -d $O/chan_stall -o chan -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $O/chan_stall.log 2>&1
find $O -name "*counter_collection.csv" | head; for f in $(find $O -name "*counter_collection.csv"); do head -3 $f; grep -c march $f; done
tail -3 $O/pytest.log; cat $O/vmm.jsonl | tail -12; cat $O/size_sweep.md | head -12
