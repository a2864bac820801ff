#!/bin/bash
# profiles/r06_rows4098_nt.sh -- GPU box (one bounded experiment): the non-temporal policy on WRF's own
# 4098-element rows.  AMT_NT_LOAD 1 (t, ft, ww_1 non-temporal) was tuned on padded rows, where no line is shared between tiles; on
# 4098-element rows the edge line of exactly those streams IS shared with the neighbouring tile and nt asks L2 to drop it.
# A/B libraries (make -C csrc variant ...): nt0 = no nt loads, nt2 = u, u_1 nt as well, nts1 = nt stores, nt0nts1 = both changes.
# Each run is bench.py with its same-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes (which inherit AMT_LIBRARY).
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_rows4098_nt; mkdir -p $O
D=wrf-model-cuda-sample_amd/csrc/build/diag
for ROUND in 1 2; do
for V in default nt0 nt2 nts1 nt0nts1; do
  for AL in 1 32; do
    if [ $V = default ]; then unset AMT_LIBRARY; else export AMT_LIBRARY=$PWD/$D/libamt_$V.so; fi
    python3 bench.py --align-elems $AL --no-cpu-baseline --no-box-probe --steps 10 --warmup 3 > $O/${V}_al${AL}_r$ROUND.json 2> $O/${V}_al${AL}_r$ROUND.err
  done
done
done
unset AMT_LIBRARY
python3 - $O <<'PY'
import json, sys, glob
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        r = d["roofline"]
        rows.append((f.split("/")[-1][:-5], d["ms_per_step_median"], r["frac"], r.get("traffic_over_algorithmic"),
                     r.get("traffic_read_bytes"), r.get("traffic_write_bytes"), d["config"]["placement_probe_ms"], d["verified_vs_oracle"]))
    except Exception as e:
        rows.append((f, "failed", str(e)))
for r in rows:
    print(*r)
PY
