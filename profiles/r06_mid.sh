#!/bin/bash
# r06 mid-round pass (GPU box): the cache-policy selection (NTL) against the parity suites and a 20 000-case random campaign, the
# new bench line at N = 1 (layout keys, wrf_rows, overlapped CPU prebuild: wall clock), the first-contact ladder with two ranks
# sharing the GPU, and WRF's unpadded rows with the policy as the launcher now picks it / forced the old way.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_mid; mkdir -p $O
rm -rf oracle/_native
( time python3 bench.py ) > $O/bench_noargs.json 2> $O/bench_noargs.err
( time timeout 1500 python3 -m pytest tests/test_gpu_15_stream_policy.py tests/test_gpu_23_oneshot_devices.py tests/test_gpu_20_host_cache.py tests/test_gpu_21_fortran_host.py tests/test_gpu_22_dropin_linkswap.py tests/test_gpu_00_configs.py tests/test_gpu_10_parity.py tests/test_gpu_11_shapes.py tests/test_gpu_14_tall.py -x -q -m gpu ) > $O/pytest_parity.log 2>&1; echo "rc $?" >> $O/pytest_parity.log
( time AMT_RANDOM_CASES=20000 AMT_RANDOM_SEED=606 timeout 1500 python3 -m pytest tests/test_gpu_12_random.py -x -q -m gpu ) > $O/pytest_campaign.log 2>&1; echo "rc $?" >> $O/pytest_campaign.log
cp gpurun_out/random_campaign_20000.json $O/ 2>/dev/null
( time timeout 1500 python3 -m pytest tests/test_gpu_90_bench_multirank.py -x -q -m gpu ) > $O/pytest_bench.log 2>&1; echo "rc $?" >> $O/pytest_bench.log
for R in 1 2 3; do
  python3 bench.py --align-elems 1 --no-cpu-baseline --no-box-probe --steps 10 --warmup 3 --wrf-rows-steps 0 > $O/rows4098_auto_r$R.json 2> $O/rows4098_auto_r$R.err
  AMT_MARCH_NT=1 python3 bench.py --align-elems 1 --no-cpu-baseline --no-box-probe --steps 10 --warmup 3 --wrf-rows-steps 0 > $O/rows4098_nt1_r$R.json 2> $O/rows4098_nt1_r$R.err
done
( time python3 bench.py --gpus 2 --share-gpu --steps 10 --warmup 3 --no-box-probe ) > $O/bench_share2_both.json 2> $O/bench_share2_both.err
tail -3 $O/pytest_parity.log $O/pytest_campaign.log $O/pytest_bench.log
grep real $O/bench_noargs.err
python3 - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(f.split("/")[-1], "no line", e); continue
    r = d.get("roofline", {})
    print(f.split("/")[-1], d.get("ms_per_step"), d.get("ms_per_step_median"), r.get("frac"), r.get("traffic_over_algorithmic"),
          (d.get("config") or {}).get("kernel", "")[-30:], (d.get("config") or {}).get("placement_probe_ms"), d.get("wrf_rows"), d.get("value_transport"))
PY
