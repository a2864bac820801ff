"""In-process A/B of wave shapes of the march kernel on the SAME resident arrays (placement is then
common to all; interleaved rounds, median and min):
  python profiles/ab_shapes.py --dtype f64 --ni 4096 --nk 80 --nj 2048 auto 1,4,2,0,1 1,4,4,0,1 ...
a shape is vw,kpt,hl,xd,dma[,jrows[,maxwaves]]; `auto` = the launcher's own choice; `column` = the column kernel;
`tN` = the launcher's shape with the block schedule N (t0: uniform blocks, t1: the launcher's tapering rule, t48: tapered,
longest block 48 rows).
Prints one line per shape: median / min ms, algorithmic TB/s and the fraction of 8 TB/s."""
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
ap.add_argument("--inner", type=int, default=5)
ap.add_argument("--unaligned", action="store_true")
ap.add_argument("shapes", nargs="+")
a = ap.parse_args()
pkg = g.load_package()
L = pkg.load_library()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=not a.unaligned)
dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
abytes = np.dtype(dtype).itemsize * a.ni * a.nj * (11 * a.nk + 14)


def setup(spec):
    if spec == "auto":
        L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
        return 0
    if spec == "column":
        L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
        return 1
    v = [int(x) for x in spec.split(",")]
    v += [0] * (7 - len(v))
    L.amt_march_force_shape(*v)
    return 2


times, names = {s: [] for s in a.shapes}, {}
for rnd in range(a.rounds):
    for spec in a.shapes:
        variant = setup(spec)
        try:
            call = pkg.bind_device_call(*dev.args(), variant=variant)
            call()
        except pkg.AmtError as e:
            names[spec] = f"cannot run: {e}"
            continue
        torch.cuda.synchronize()
        names[spec] = L.amt_march_last_kernel().decode() if variant != 1 else "amt_column_kernel"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.inner):
            call()
        e1.record()
        torch.cuda.synchronize()
        times[spec].append(e0.elapsed_time(e1) / a.inner)
L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
print(f"# {a.ni}x{a.nk}x{a.nj} {a.dtype} idim {b.idim}: {abytes / 1e9:.2f} GB algorithmic per sweep")
for spec in a.shapes:
    v = times[spec]
    if not v:
        print(f"{spec:>16s}: {names.get(spec)}")
        continue
    med = float(np.median(v))
    print(f"{spec:>16s}: median {med:8.3f} ms  min {min(v):8.3f}  {abytes / med / 1e9:6.3f} TB/s  {abytes / med / 1e9 / 8:.3f} of 8 TB/s   {names[spec][16:]}")
