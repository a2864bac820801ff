#!/bin/bash
# profiles/r05_rows4098.sh -- GPU box: WRF's own row length (ims:ime = 0:4097, 4098-element rows: bench.py --align-elems 1) against
# the block length: do shorter blocks (neighbouring tiles drift apart less) keep the shared edge lines in L2?
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_rows4098; mkdir -p $O
for JR in 0 512 256 128 64 32; do
  AMT_MARCH_JROWS=$JR python3 bench.py --align-elems 1 --no-cpu-baseline --no-box-probe --steps 10 > $O/jrows$JR.json 2> $O/jrows$JR.err
done
python3 - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/jrows*.json"), key=lambda s: int(s.split("jrows")[1].split(".")[0])):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], r["frac"], r.get("traffic_over_algorithmic"), d["config"]["kernel"][-40:], d["config"]["placement_probe_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
