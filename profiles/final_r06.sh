#!/bin/bash
# r06 final pass (GPU box): a COLD box (no -march=native CPU libraries), the driver's three steps on the final build, a kernel
# trace of the driver's bench command, the first-contact ladder with two ranks sharing the GPU, and the rocprofv3 summaries of the
# two layouts of the headline workload.
set -u
export TMPDIR=/tmp
O=gpurun_out/final6; mkdir -p $O
rm -rf oracle/_native
( time timeout 3300 python3 -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
rm -rf oracle/_native
( time python3 bench.py ) > $O/bench_noargs.json 2> $O/bench_noargs.err
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_traced.json 2> $O/bench_traced.err
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done; rm -rf $O/trace
( time python3 bench.py --gpus 2 --share-gpu --steps 10 --warmup 3 --no-box-probe ) > $O/bench_share2_both.json 2> $O/bench_share2_both.err
tail -n 3 $O/pytest.log; tail -n 2 $O/smoke.log; grep real $O/bench_noargs.err $O/bench_driver.err $O/bench_share2_both.err; head -4 $O/kernel_stats.csv | cut -c1-200
bash profiles/collect.sh r06_f64_4096x60x4096 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe --wrf-rows-steps 0 > $O/collect_a.log 2>&1
bash profiles/collect.sh r06_f64_4096x60x4096_rows4098 --align-elems 1 --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe --wrf-rows-steps 0 > $O/collect_b.log 2>&1
cp profiles/r06_*kernel_stats.csv profiles/r06_*pmc.json profiles/hbm_traffic.json $O/ 2>/dev/null
python3 - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(f.split("/")[-1], "no line", e); continue
    r = d.get("roofline", {})
    print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), r.get("frac"), r.get("traffic_over_algorithmic"), d.get("verified_vs_oracle"),
          (d.get("wrf_rows") or {}).get("ms_per_step"), (d.get("wrf_rows") or {}).get("traffic_over_algorithmic"), d.get("value_transport"),
          (d.get("cpu_baseline") or {}).get("value"))
PY
ls $O
