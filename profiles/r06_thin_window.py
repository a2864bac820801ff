"""What a ONE-COLUMN window costs (the boundary columns of an i x j patch, csrc/amt_grid.hip: grid_edges), by wave shape and rows
per workgroup, against the two boundary rows in one launch and the bare patch: where do the +8 % / +20 % of profiles/
r05_grid_loopback.md go, and does a narrower tile (more level groups per wave: 16 columns instead of 64) or another block length
buy them back?   python profiles/r06_thin_window.py [--ni 2048 --nj 2048]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ni", type=int, default=2048)
ap.add_argument("--nk", type=int, default=60)
ap.add_argument("--nj", type=int, default=2048)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
pkg = g.load_package()
S, L = pkg.synth, pkg.load_library()
dtype = np.float64 if a.dtype == "f64" else np.float32
gdims = (3 * a.ni, a.nk, 3 * a.nj)
pb = S.patch_bounds(S.domain_bounds(*gdims), 1, 1, 3, 3, align_elems=32)
dev = S.make_patch(pb, pkg.GridConfig(), dtype=dtype, seed=1, global_dims=gdims, device="cuda:0")


def timed(call, n):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n          # microseconds, back to back on one stream


def window(**kw):
    return pkg.bind_device_call(*dev.with_bounds(**kw).args())


bare = timed(window(), 30)
print(f"{a.ni}x{a.nk}x{a.nj} {a.dtype} patch: bare sweep {bare:.1f} us")
interior = timed(window(its=pb.its + 1, ite=pb.ite - 1, jts=pb.jts + 1, jte=pb.jte - 1), 30)
print(f"interior (window one cell in from every side): {interior:.1f} us ({100 * (interior / bare - 1):+.1f} %), kernel {L.amt_march_last_kernel().decode()}")
rows2 = timed(window(jts=pb.jts, jte=pb.jts), a.reps) + timed(window(jts=pb.jte, jte=pb.jte), a.reps)
print(f"two boundary rows as two one-row launches: {rows2:.1f} us")
for hl in (0, 1, 2, 4):
    for jrows in (0, 1, 2, 4, 8, 16, 32):
        L.amt_march_force_shape(0, 0, hl, -1, 1, jrows, 0)
        try:
            call = window(its=pb.its, ite=pb.its, jts=pb.jts + 1, jte=pb.jte - 1)
            us = timed(call, a.reps)
            print(f"  one column, hl {hl or 'auto'} jrows {jrows or 'auto':>4}: {us:7.1f} us   {L.amt_march_last_kernel().decode()[16:]}")
        except Exception as e:  # noqa: BLE001
            print(f"  one column, hl {hl} jrows {jrows}: not runnable ({type(e).__name__})")
L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
