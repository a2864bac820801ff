"""Interleaved A/B of resident i paddings (row strides) in ONE process: rounds x layouts, median/min.
usage: python profiles/ab_layout.py "32,0" "64,0" "64,96"   (align_elems,idim_extra; 3 fit in HBM)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
S = pkg.synth
layouts = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(32, 0), (64, 0)]
calls = []
for al, extra in layouts:
    b = S.domain_bounds(4096, 60, 4096, aligned=True, align_elems=al)
    b = b.replace(ime=b.ime + extra)
    dev = S.make_patch(b, pkg.GridConfig(), seed=1, device="cuda:0")
    calls.append((al, extra, b.idim, pkg.bind_device_call(*dev.args()), dev))
torch.cuda.synchronize()
times = {c[:3]: [] for c in calls}
for rnd in range(8):
    for al, extra, idim, call, _ in calls:
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        times[(al, extra, idim)].append(e0.elapsed_time(e1) / 5)
for k, v in times.items():
    print(f"align {k[0]:3d} extra {k[1]:4d} idim {k[2]:5d}: median {np.median(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}")
