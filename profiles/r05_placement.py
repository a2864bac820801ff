"""profiles/r05_placement.py -- why does one allocation of an array stream faster than another?  

K allocations of ONE 8 GiB array in one process (the allocator's cache emptied and a spacer of another size in front each
time: fresh physical pages), each read three times by the read-only streaming kernel of the library (amt_calib_stream_rate mode
1: 16 B per lane, nothing written) -- the placement effect shows in a plain read stream (profiles/r03_placement.md section 1b), so
the kernel under study is the simplest one that has it.  Run plain for the rates, and under
  rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -o p -- python3 profiles/r05_placement.py --k 10
for the same dispatches with hardware counters; `--report <dir>` then joins duration and counters per dispatch and prints the
correlation over the placements (the dispatches of one process: durations under the profiler are serialized, relative only)."""
import argparse
import csv
import ctypes
import glob
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def measure(a):
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    L = pkg.load_library()
    torch.cuda.set_device(0)
    nbytes = a.gib << 30
    sink = torch.zeros(8, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    h = ctypes.c_void_p(stream.cuda_stream)
    out = []
    for k in range(a.k):
        spacer = torch.empty((k * 1237 + 311) << 20, dtype=torch.uint8, device="cuda")
        buf = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
        del spacer
        buf.fill_(1.0)
        torch.cuda.synchronize()
        rates = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pkg.lib.check(L.amt_calib_stream_rate(h, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(nbytes), 1))
            e1.record()
            torch.cuda.synchronize()
            rates.append(nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        out.append({"placement": k, "address": hex(buf.data_ptr()), "GBps": [round(x, 1) for x in rates]})
        print(json.dumps(out[-1]), flush=True)
        del buf
        torch.cuda.empty_cache()
    best = [max(r["GBps"]) for r in out]
    print(json.dumps({"summary": "best of reps per placement", "min": min(best), "max": max(best), "spread_pct": round(100 * (max(best) / min(best) - 1), 2)}))


def report(d):
    disp = {}
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "amt_stream_read_kernel" in r["Kernel_Name"]:
                disp[r["Dispatch_Id"]] = {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}
    names = set()
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Dispatch_Id"] in disp:
                disp[r["Dispatch_Id"]][r["Counter_Name"]] = disp[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names.add(r["Counter_Name"])
    rows = [disp[k] for k in sorted(disp, key=int)]
    if not rows:
        print("no dispatches found in", d)
        return
    ns = np.array([r["ns"] for r in rows], float)
    print(f"{len(rows)} dispatches; duration us min {ns.min() / 1e3:.1f} max {ns.max() / 1e3:.1f} (spread {100 * (ns.max() / ns.min() - 1):.1f} %)")
    for n in sorted(names):
        v = np.array([r.get(n, np.nan) for r in rows], float)
        ok = np.isfinite(v)
        if ok.sum() < 3 or v[ok].std() == 0:
            print(f"  {n:45s} constant {v[ok][0] if ok.any() else float('nan'):.4g}")
            continue
        c = np.corrcoef(ns[ok], v[ok])[0, 1]
        print(f"  {n:45s} min {v[ok].min():.4g} max {v[ok].max():.4g} (spread {100 * (v[ok].max() / max(v[ok].min(), 1e-30) - 1):.1f} %)  corr with duration {c:+.2f}")
    print("  per dispatch: us, " + ", ".join(sorted(names)))
    for r in rows:
        print("   ", f"{r['ns'] / 1e3:9.1f}", *[f"{r.get(n, float('nan')):.5g}" for n in sorted(names)])


ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--gib", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--report", default="")
a = ap.parse_args()
if a.report:
    report(a.report)
else:
    measure(a)
