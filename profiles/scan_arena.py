"""Scan the stagger between consecutive 3-D arrays carved out of one arena (one process, one
allocation reused): sweep time as a function of the relative placement of the ten streams.
usage: python profiles/scan_arena.py [--ni 4096 --nk 60 --nj 4096 --dtype f64] s0 s1 ...  (bytes)"""
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
ap.add_argument("staggers", type=int, nargs="*")
a = ap.parse_args()
pkg = g.load_package()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
base = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
order = list(S.RANK3) + [n for n in S.FIELD_NAMES if n not in S.RANK3]
sizes = {n: base.arrays[n].numel() * base.arrays[n].element_size() for n in order}
smax = max(a.staggers) if a.staggers else 0
total = sum((sz + 255) // 256 * 256 + smax for sz in sizes.values()) + (1 << 21)
arena = torch.empty(total, dtype=torch.uint8, device="cuda:0")
print(f"array bytes {sizes['t']}  arena base mod 2MiB {arena.data_ptr() % (1 << 21)}", flush=True)


def timed(call):
    call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    return best


print(f"separate allocations: {timed(pkg.bind_device_call(*base.args())):.3f} ms", flush=True)
for stagger in a.staggers:
    off = (-arena.data_ptr()) % (1 << 21)
    arrays = {}
    for n in order:
        src = base.arrays[n]
        view = arena[off:off + sizes[n]].view(src.dtype).view(src.shape)
        view.copy_(src)
        arrays[n] = view
        off += (sizes[n] + 255) // 256 * 256 + (stagger if n in S.RANK3 else 0)
    dev = S.Patch(base.bounds, base.config, arrays, base.rdx, base.rdy, base.dts, base.epssm, base.global_dims)
    t = timed(pkg.bind_device_call(*dev.args()))
    print(f"stagger {stagger:9d}  pitch mod 4096 {(sizes['t'] + stagger) % 4096:5d}  mod 64Ki {(sizes['t'] + stagger) % 65536:6d}: {t:.3f} ms", flush=True)
