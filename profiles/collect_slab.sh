#!/bin/bash
# profiles/collect_slab.sh -- GPU box: HBM traffic of ONE RANK's slab sweep at N = 2, 4, 8 (rows 2048, 1024, 512 of the
# 4096x60x4096 fp64 domain) through the native stepper with RCCL in loopback (profiles/slab_loopback.py): FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes, summed over the advance_mu_t kernels of a run and divided by its sweeps.
# Refreshes profiles/hbm_traffic.json entries 4096x60x4096_f64_n{2,4,8} (what bench.py reports as roofline.traffic
# for N > 1: per rank and sweep; recorded, not re-measured in the run).
set -u
export TMPDIR=/tmp
O=gpurun_out/prof_slab; mkdir -p $O
SW=20
for NJ in 2048 1024 512; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --kernel-trace --pmc $C -d $O/nj${NJ}_$C -o pmc -- python3 profiles/slab_loopback.py --nj $NJ --sweeps $SW > $O/nj${NJ}_$C.log 2>&1
  done
done
python3 - "$O" "$SW" <<'PY'
import csv, glob, json, os, sys
O, SW = sys.argv[1], int(sys.argv[2])
here = "profiles"
table = json.load(open(f"{here}/hbm_traffic.json"))
for nj, n in ((2048, 2), (1024, 4), (512, 8)):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        s, launches = 0.0, 0
        for f in glob.glob(f"{O}/nj{nj}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "amt_march" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    s += float(r["Counter_Value"]); launches += 1
        tot[c] = (s, launches)
    sweeps = 3 * (SW + 3)                     # bare, overlap, no-overlap: SW timed + 3 warm-up sweeps each
    fetch, write = tot["FETCH_SIZE"][0] / sweeps, tot["WRITE_SIZE"][0] / sweeps
    rec = {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "read_bytes": int(2 * fetch * 1024), "write_bytes": int(write * 1024),
           "per": "rank and sweep (interior launch + edge launch of one j-slab)",
           "source": f"profiles/collect_slab.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH doubled per the gfx950 calibration) "
                     f"over the advance_mu_t kernels of profiles/slab_loopback.py --nj {nj} (native stepper, RCCL loopback), "
                     f"{tot['FETCH_SIZE'][1]} launches / {sweeps} sweeps",
           "algorithmic_bytes": 8 * 4096 * nj * (11 * 60 + 14)}
    table[f"4096x60x4096_f64_n{n}"] = rec                      # what bench.py looks up: the whole domain's dims and N
    table[f"4096x60x{nj}_f64_n{n}"] = dict(rec, alias_of=f"4096x60x4096_f64_n{n}")    # the same record under the slab's own dims
    print(n, rec["hbm_bytes_per_launch"], rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes"])
json.dump(table, open(f"{here}/hbm_traffic.json", "w"), indent=1)
PY
cp profiles/hbm_traffic.json $O/hbm_traffic.json
cp profiles/hbm_traffic.json $O/hbm_traffic.json
