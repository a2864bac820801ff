"""profiles/summarize.py TAG RAW_DIR -- condense raw rocprofv3 CSVs into the committed summaries."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, raw = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))

# VGPRs, scratch and static LDS of the march kernels come from the compiler's own resource remarks
# (csrc/build/march_resources.txt, `make -C csrc resources`), dynamic LDS and the launch geometry from the
# launcher's AMT_MARCH_VERBOSE line: rocprofv3's kernel-trace CSV reports VGPR_Count 64 and LDS 0 for
# these kernels, which contradicts both.
sys.path.insert(0, os.path.join(here, "..", "wrf-model-cuda-sample_amd", "tools"))
try:
    import kernel_resources
    if not kernel_resources.REMARKS.exists():
        import subprocess
        subprocess.run(["make", "-C", str(kernel_resources.CSRC), "resources"], check=True, capture_output=True)
    RES = {f"{r['kernel']}<{r['targs']}>": r for r in kernel_resources.parse()}
except Exception as e:                                   # noqa: BLE001
    print("no compiler resource table:", e)
    RES = {}
PLAN = {}                                                # kernel name -> (waves, dynamic LDS bytes, tiles, blocks, rows)
for name in ("bench_plain", "bench_trace"):
    path = os.path.join(raw, name + ".log")
    if os.path.exists(path):
        import re
        for line in open(path, errors="replace"):
            m = re.search(r"\[amt march\].*-> (amt_march_kernel<[^>]*>) (FULL|ragged): (\d+) waves, (\d+) B LDS, (\d+) tiles x (\d+) blocks of (\d+) rows", line)
            if m:
                full = "true" if m.group(2) == "FULL" else "false"
                # the launcher's label -> the demangled name rocprofv3 prints (FULL / ragged and the cache policy are template arguments)
                PLAN[m.group(1).replace("FULL", full).replace(", nt>", ", 1>").replace(", cached>", ", 0>")] = tuple(int(x) for x in m.groups()[2:])


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(raw, pattern), recursive=True):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def short(name):
    return name.split("(")[0].replace("void ", "")[:80]


# ---- pass 1: kernel durations ------------------------------------------------------
stats = rows("trace/**/*kernel_stats.csv")
trace = rows("trace/**/*kernel_trace.csv")
dur = defaultdict(list)
for r in trace:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["kernel,calls,total_us,avg_us,min_us,max_us,vgpr_compiler,scratch_bytes_per_lane,sgpr_compiler,"
         "lds_dynamic_bytes,waves_per_workgroup,workgroup,grid,rows_per_workgroup"]
meta = {}
for r in trace:
    meta[short(r["Kernel_Name"])] = (r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("LDS_Block_Size", ""),
                                     r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("Grid_Size", r.get("Grid_Size_X", "")))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    m = meta.get(k, ("",) * 5)
    r = RES.get(k, {})
    pl = PLAN.get(k, ("", "", "", "", ""))
    lines.append(f"\"{k}\",{len(v)},{sum(v):.1f},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f},"
                 f"{r.get('vgprs', '')},{r.get('scratch_bytes_per_lane', '')},{r.get('sgprs', '')},"
                 f"{pl[1]},{pl[0]},{m[3]},{m[4]},{pl[4]}")
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

# ---- roofline.traffic of bench.py: corrected HBM bytes per launch of the dominant kernel ----------
line = out.get("bench_plain") or out.get("bench_trace")
dom = max((k for k in dur if "amt_" in k and "synth" not in k and "calib" not in k), key=lambda k: sum(dur[k]), default=None)
if line and dom and dom in out["kernels"] and "FETCH_SIZE" in out["kernels"][dom] and "WRITE_SIZE" in out["kernels"][dom]:
    cfg = line["config"]
    key = f"{cfg['ni']}x{cfg['nk']}x{cfg['nj']}_{line['dtype']}_n{line['n_gpus']}"
    if cfg.get("aligned") is False:                      # WRF's own unpadded rows (bench.py --align-elems 1): a key of its own
        key += f"_rows{cfg['idim']}"
    fetch = out["kernels"][dom]["FETCH_SIZE"]["mean_per_launch"]
    write = out["kernels"][dom]["WRITE_SIZE"]["mean_per_launch"]
    path = os.path.join(here, "hbm_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[key] = {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "read_bytes": int(2 * fetch * 1024),
                  "write_bytes": int(write * 1024),
                  "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE "
                            f"doubled per the gfx950 calibration in profiles/README.md)",
                  "kernel": dom, "kernel_avg_us": round(sum(dur[dom]) / len(dur[dom]), 1),
                  "algorithmic_bytes": line["roofline"]["algorithmic_bytes_per_launch"]}
    if cfg.get("aligned") is False:
        table[key]["layout"] = f"WRF's own unpadded rows (bench.py --align-elems 1: {cfg['idim']}-element rows)"
    json.dump(table, open(path, "w"), indent=1)
    print("hbm_traffic.json:", key, table[key])
print(open(os.path.join(here, f"{tag}_kernel_stats.csv")).read())
print(json.dumps(out["kernels"], indent=1)[:3000])
print(json.dumps(out["calibration"], indent=1)[:3000])
