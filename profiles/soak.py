"""Determinism soak: the same sweep from the same inputs N times; every output's bit-pattern
checksum must be identical every time (a race in the barrier / LDS-DMA schedule would show as a
rare difference).  python profiles/soak.py [--n 2000] [--dtype f64] [--ni 4096 --nk 60 --nj 512]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--ni", type=int, default=4096)
ap.add_argument("--nk", type=int, default=60)
ap.add_argument("--nj", type=int, default=512)
ap.add_argument("--dtype", default="f64")
a = ap.parse_args()
pkg = g.load_package()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
ibits = torch.int64 if a.dtype == "f64" else torch.int32
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
dev = S.make_patch(b, pkg.GridConfig(specified=True), dtype=dtype, seed=3, device="cuda:0")
inout = ("ww", "t", "mu")
pristine = {n: dev.arrays[n].clone() for n in inout}
call = pkg.bind_device_call(*dev.args())


def checksum():
    return tuple(int(dev.arrays[n].view(ibits).sum(dtype=torch.int64).item()) for n in S.OUTPUTS)


t0 = time.time()
first = None
bad = 0
for it in range(a.n):
    for n in inout:
        dev.arrays[n].copy_(pristine[n])
    call()
    c = checksum()
    if first is None:
        first = c
    elif c != first:
        bad += 1
        print(f"iteration {it}: checksums differ: {c} vs {first}", flush=True)
print(f"{a.n} sweeps of {a.ni}x{a.nk}x{a.nj} {a.dtype}: {bad} differing, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
