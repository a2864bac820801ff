"""In-process A/B of several builds of the library on the SAME resident arrays (placement is
then common to all): python profiles/ab_libs.py [--dtype f64 --ni 4096 --nk 60 --nj 4096] a.so b.so:ENV=1,ENV2=3 ...
(a spec may carry environment knobs, set around that variant's launches)"""
import os
import argparse
import ctypes
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
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
pkg = g.load_package()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
args = dev.args()
arrays = [x for x in args if hasattr(x, "data_ptr")]
scal = [x for x in args if isinstance(x, float)]
ints = [x for x in args if isinstance(x, int) and not isinstance(x, bool)]
assert len(arrays) == 26 and len(scal) == 4 and len(ints) == 17, (len(arrays), len(scal), len(ints))
P, F = ctypes.c_void_p, (ctypes.c_double if a.dtype == "f64" else ctypes.c_float)
calls = []
for spec in a.libs:
    path, _, envs = spec.partition(":")
    env = dict(kv.split("=") for kv in envs.split(",") if kv)
    L = ctypes.CDLL(str(Path(path).resolve()))
    fn = getattr(L, "amt_advance_mu_t_device_" + a.dtype)
    fn.restype = ctypes.c_int
    fn.argtypes = [P, ctypes.c_int] + [P] * 18 + [F] * 4 + [P] * 8 + [ctypes.c_int] * 20
    cargs = [P(torch.cuda.current_stream().cuda_stream), 0] + [P(x.data_ptr()) for x in arrays[:18]] + scal \
        + [P(x.data_ptr()) for x in arrays[18:]] + [0, 0, 0] + ints

    def call(fn=fn, cargs=cargs, env=env):
        for k, v in env.items():
            os.environ[k] = v
        rc = fn(*cargs)
        for k in env:
            os.environ.pop(k, None)
        assert rc == 0, rc
    calls.append((Path(path).stem + (":" + envs if envs else ""), call))
times = {n: [] for n, _ in calls}
for rnd in range(a.rounds):
    for n, call in calls:
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 5)
for n, v in times.items():
    print(f"{n:>40s}: median {np.median(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}")
