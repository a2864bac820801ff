"""Phase timing inside the march kernel (instrumentation build, never the product):

    make -C wrf-model-cuda-sample_amd/csrc variant NAME=stamps DEFS=-DAMT_STAMPS=1
    AMT_LIBRARY=wrf-model-cuda-sample_amd/csrc/build/diag/libamt_stamps.so python profiles/stamps.py --dtype f32 --ni 8192 --nk 80 --nj 512

The middle workgroup of the launch records s_memtime at eight points of rows 4..35 of its j block, in its first cell
wave, its last cell wave and its column wave.  Printed: mean cycles between consecutive points (and the share of a
row), per wave.  Points of a cell wave: 0 row start | 1 P1 done (loads waited for, AB written) | 2 past barrier 1 |
3 DMA + P3 loads issued | 4 past barrier 2 | 5 increments written | 6 past barriers 3+4 | 7 P3 done (stores issued).
Column wave: 0 row start | 1 2-D loads issued | 2 past barrier 1 | 3 chain 1 done | 4 past barrier 2 | 5 mass update
stores issued | 6 past barrier 3 | 7 chain 2 done."""
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
ap.add_argument("--nj", type=int, default=1024)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--shape", default="", help="vw,kpt,hl,xd,dma[,jrows[,maxwaves]] to force")
a = ap.parse_args()
pkg = g.load_package()
L = pkg.load_library()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
if a.shape:
    v = [int(x) for x in a.shape.split(",")]
    v += [0] * (7 - len(v))
    L.amt_march_force_shape(*v)
call = pkg.bind_device_call(*dev.args(), variant=2)
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
buf = np.zeros((3, 32, 8), dtype=np.uint64)
L.amt_diag_stamps.restype = ctypes.c_int
L.amt_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = L.amt_diag_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0, rc
label = L.amt_march_last_kernel().decode()
print(f"# {a.ni}x{a.nk}x{a.nj} {a.dtype}: {ms:.3f} ms per sweep; {label}")
t = buf.astype(np.int64)
row = np.diff(t[:, :, 0], axis=1)                       # row start to next row start
print(f"# cycles per row (s_memtime units): cell0 {row[0].mean():.0f}  cellN {row[1].mean():.0f}  column {row[2].mean():.0f}")
names = ["cell wave 0", "last cell wave", "column wave"]
for w in range(3):
    seg = np.diff(t[w], axis=1)[:-1]                    # [row][7]: between the eight points
    tail = t[w, 1:, 0] - t[w, :-1, 7]                   # last point to the next row's start
    parts = np.concatenate([seg, tail[:, None]], axis=1)
    mean = parts.mean(axis=0)
    tot = mean.sum()
    print(f"{names[w]:>15s}: " + "  ".join(f"{i}->{(i + 1) % 8}: {m:6.0f} ({100 * m / tot:4.1f}%)" for i, m in enumerate(mean)))
