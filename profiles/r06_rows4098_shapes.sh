#!/bin/bash
# profiles/r06_rows4098_shapes.sh -- GPU box: is another EXISTING instantiation better on WRF's unpadded 4098-element rows than the one
# the launcher picks (<double,1,4,1,0,DMA,16>, plain once-read loads)?  Launcher-only question: shapes forced through AMT_MARCH_*.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_rows4098_shapes; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 bench.py --align-elems 1 --no-cpu-baseline --no-box-probe --steps 10 --warmup 3 --wrf-rows-steps 0 > $O/$tag.json 2> $O/$tag.err; }
for R in 1 2; do
  run default_r$R AMT_X=1
  run xd1_r$R AMT_MARCH_XD=1
  run reg_r$R AMT_MARCH_DMA=0
  run hl2_r$R AMT_MARCH_HL=2
  run hl2k4_r$R AMT_MARCH_HL=2 AMT_MARCH_KPT=4
done
python3 - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], d["ms_per_step_median"], r["frac"], r.get("traffic_over_algorithmic"), d["config"]["kernel"][16:70], d["config"]["placement_probe_ms"], d["verified_vs_oracle"])
    except Exception as e:
        print(f, "failed", e)
PY
