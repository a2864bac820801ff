"""profiles/summarize.py TAG RAW_DIR -- condense raw rocprofv3 CSVs into the committed summaries."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, raw = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(raw, pattern), recursive=True):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


# ---- pass 1: kernel durations ------------------------------------------------------
stats = rows("trace/**/*kernel_stats.csv")
trace = rows("trace/**/*kernel_trace.csv")
dur = defaultdict(list)
for r in trace:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["kernel,calls,total_us,avg_us,min_us,max_us,vgpr,sgpr,lds_bytes,workgroup,grid"]
meta = {}
for r in trace:
    meta[short(r["Kernel_Name"])] = (r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("LDS_Block_Size", ""),
                                     r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("Grid_Size", r.get("Grid_Size_X", "")))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    m = meta.get(k, ("",) * 5)
    lines.append(f"{k},{len(v)},{sum(v):.1f},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f},{m[0]},{m[1]},{m[2]},{m[3]},{m[4]}")
open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w").write("\n".join(lines) + "\n")


# ---- PMC passes --------------------------------------------------------------------
def pmc(subdir):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(f"{subdir}/**/*counter_collection.csv"):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


out = {"tag": tag, "kernels": {}, "calibration": {}}
for sub in ("fetch", "write", "l2"):
    for k, cs in pmc(sub).items():
        for c, v in cs.items():
            out["kernels"].setdefault(k, {})[c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
for sub in ("calib_fetch", "calib_write"):
    for k, cs in pmc(sub).items():
        for c, v in cs.items():
            out["calibration"].setdefault(k, {})[c] = {"launches": len(v), "values": v}
for name in ("bench_plain", "bench_trace", "bench_fetch"):
    p = os.path.join(raw, name + ".log")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                out[name] = json.loads(line)
json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
print(open(os.path.join(here, f"{tag}_kernel_stats.csv")).read())
print(json.dumps(out["kernels"], indent=1)[:3000])
print(json.dumps(out["calibration"], indent=1)[:3000])
