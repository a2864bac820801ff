"""A/B in ONE process: the 26 resident arrays as separate allocations vs carved out of one arena
with a chosen stagger between consecutive arrays (does the relative placement of the ten 3-D
streams matter?).  usage: python profiles/ab_arena.py sep 0 4096 1052672 ...  (stagger bytes)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
S = pkg.synth
b = S.domain_bounds(4096, 60, 4096, aligned=True)
base = S.make_patch(b, pkg.GridConfig(), seed=1, device="cuda:0")
variants = sys.argv[1:] or ["sep", "0", "4096"]
calls = []
for v in variants:
    if v == "sep":
        dev = base
    else:
        stagger = int(v)
        sizes = {n: base.arrays[n].numel() * base.arrays[n].element_size() for n in S.FIELD_NAMES}
        total = sum((sz + 255) // 256 * 256 + stagger for sz in sizes.values()) + 4096
        arena = torch.empty(total, dtype=torch.uint8, device="cuda:0")
        off = (-arena.data_ptr()) % 4096
        arrays = {}
        for n in S.FIELD_NAMES:
            src = base.arrays[n]
            view = arena[off:off + sizes[n]].view(src.dtype).view(src.shape)
            view.copy_(src)
            arrays[n] = view
            off += (sizes[n] + 255) // 256 * 256 + stagger
        dev = S.Patch(base.bounds, base.config, arrays, base.rdx, base.rdy, base.dts, base.epssm, base.global_dims)
    calls.append((v, pkg.bind_device_call(*dev.args()), dev))
torch.cuda.synchronize()
times = {c[0]: [] for c in calls}
for rnd in range(6):
    for v, call, _ in calls:
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 5)
for k, v in times.items():
    print(f"{k:>10s}: median {np.median(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}")
