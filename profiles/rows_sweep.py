"""Rows per workgroup, swept in one process on the same resident arrays (amt_march_force_shape):
python profiles/rows_sweep.py --dtype f64 --ni 4096 --nk 60 --nj 4096 --rows 0,64,128,256,512,1024
(0 = the launcher's own choice).  Prints the instantiation label and the median launch time."""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ni", type=int, default=4096)
ap.add_argument("--nk", type=int, default=60)
ap.add_argument("--nj", type=int, default=4096)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--rows", default="0,64,128,256,512,1024")
ap.add_argument("--xchunk", default="0", help="comma list; ids an XCD takes per round (0: one run per launch)")
a = ap.parse_args()
pkg = g.load_package()
L = pkg.load_library()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
args = dev.args()
import ctypes
L.amt_march_set_xchunk.restype = ctypes.c_int
L.amt_march_set_xchunk.argtypes = [ctypes.c_int]
rows = [(int(x), int(c)) for c in a.xchunk.split(",") for x in a.rows.split(",")]
times = {r: [] for r in rows}
labels = {}
for rnd in range(a.rounds):
    for r in rows:
        L.amt_march_force_shape(0, 0, 0, -1, 1, r[0], 0)
        L.amt_march_set_xchunk(r[1])
        pkg.advance_mu_t(*args)
        torch.cuda.synchronize()
        labels[r] = L.amt_march_last_kernel().decode()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            pkg.advance_mu_t(*args)
        e1.record()
        torch.cuda.synchronize()
        times[r].append(e0.elapsed_time(e1) / a.reps)
L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
L.amt_march_set_xchunk(0)
print(f"{a.dtype} {a.ni}x{a.nk}x{a.nj}")
base = float(np.median(times[rows[0]]))
for r in rows:
    m = float(np.median(times[r]))
    print(f"  rows {r[0]:5d} xchunk {r[1]:3d}: median {m:8.3f} ms ({(m / base - 1) * 100:+5.2f} %)  min {min(times[r]):8.3f}  {labels[r][:110]}")
