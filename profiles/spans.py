"""When and where every workgroup of one march launch ran (instrumentation build, never the product):

    make -C wrf-model-cuda-sample_amd/csrc variant NAME=spans DEFS=-DAMT_STAMPS=1
    AMT_LIBRARY=wrf-model-cuda-sample_amd/csrc/build/diag/libamt_spans.so python profiles/spans.py --dtype f64 --rows 64 --xchunk 0

Each workgroup records REFCLK (100 MHz, common to the chip) when its first wave starts and when that wave has left its last
row, plus HW_ID / XCC_ID.  Printed: the launch's span, the spread of block durations, the time the LAST block ends after the
median end of the last round (the tail), blocks per CU, and the duration per row against the block's place in the launch."""
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
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--xchunk", type=int, default=0)
ap.add_argument("--dump", default="")
a = ap.parse_args()
pkg = g.load_package()
L = pkg.load_library()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
L.amt_march_force_shape(0, 0, 0, -1, 1, a.rows, 0)
L.amt_march_set_xchunk.argtypes = [ctypes.c_int]
L.amt_march_set_xchunk(a.xchunk)
call = pkg.bind_device_call(*dev.args(), variant=2)
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
buf = np.zeros((16384, 4), dtype=np.uint64)
L.amt_diag_spans.restype = ctypes.c_int
L.amt_diag_spans.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.amt_diag_spans(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
label = L.amt_march_last_kernel().decode()
import re
jr = int(re.search(r"jrows=(\d+)", label).group(1))
tc = {"f64": 64, "f32": 128}[a.dtype]
n = int((buf[:, 1] > 0).sum())
t = buf[:n].astype(np.int64)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0            # microseconds
dur = en - st
hw, xcc = t[:, 2], t[:, 3] & 0xF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
place = xcc * 1000 + se * 100 + sh * 10 + cu
uniq, cnt = np.unique(place, return_counts=True)
print(f"# {a.ni}x{a.nk}x{a.nj} {a.dtype} rows {a.rows} xchunk {a.xchunk}: {ms:.3f} ms; {label}")
print(f"blocks {n}; places seen {len(uniq)}; blocks per place min {cnt.min()} max {cnt.max()}")
print(f"launch span (first start .. last end) {en.max():.1f} us; starts of the first {min(n, 256)} blocks spread over {np.sort(st)[min(n, 256) - 1]:.1f} us")
print(f"block duration: median {np.median(dur):.1f} us  min {dur.min():.1f}  max {dur.max():.1f}  (p5 {np.percentile(dur, 5):.1f}  p95 {np.percentile(dur, 95):.1f})")
ends = np.sort(en)
print(f"ends of the last 256 blocks: first {ends[-min(n, 256)]:.1f}  median {np.median(ends[-min(n, 256):]):.1f}  last {ends[-1]:.1f} us  -> tail {ends[-1] - np.median(ends[-min(n, 256):]):.1f} us")
# per-XCD
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  xcc {x}: blocks {int(m.sum()):5d}  median block {np.median(dur[m]):8.1f} us  last end {en[m].max():9.1f} us  places {len(np.unique(place[m]))}")
# idle gaps per place: time between a block's end and the next block's start on the same place
gaps = []
for pl in uniq:
    m = np.where(place == pl)[0]
    o = m[np.argsort(st[m])]
    gaps += list(st[o][1:] - en[o][:-1])
if gaps:
    gaps = np.array(gaps)
    print(f"gap between consecutive blocks on a CU: median {np.median(gaps):.1f} us  p95 {np.percentile(gaps, 95):.1f}  max {gaps.max():.1f}  sum/CU {gaps.sum() / len(uniq):.1f} us")
busy = np.array([dur[place == pl].sum() for pl in uniq])
print(f"busy time per CU: median {np.median(busy):.1f}  min {busy.min():.1f}  max {busy.max():.1f} us (launch {en.max():.1f})")
if a.dump:
    np.save(a.dump, np.stack([st, en, place.astype(np.float64)], axis=1))
