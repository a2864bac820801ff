"""One rank of N's j-slab on one GPU through the NATIVE stepper with RCCL in loopback (the rank is
its own neighbour): what the exchange + the edge launches cost against the bare slab sweep.
python profiles/slab_loopback.py [--nj 512] [--dtype f64]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ni", type=int, default=4096)
ap.add_argument("--nk", type=int, default=60)
ap.add_argument("--nj", type=int, default=512)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--sweeps", type=int, default=200)
a = ap.parse_args()
pkg = g.load_package()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
gdims = (a.ni, a.nk, 3 * a.nj)
gb = S.domain_bounds(*gdims, aligned=True)
b = S.slab_bounds(gb, 1, 3)                       # a middle slab: neighbours on both sides


def timed(fn, n):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, global_dims=gdims, device="cuda:0")
bare = pkg.bind_device_call(*dev.args())
print(f"{a.ni}x{a.nk}x{a.nj} {a.dtype} slab, bare sweep (one launch): {timed(lambda n: [bare() for _ in range(n)], a.sweeps):.4f} ms")
for overlap in (True, False):
    st = pkg.patch.NativeSlabStepper(dev, 0, 1, pkg.patch.NativeSlabStepper.comm_unique_id(), loopback=True, overlap=overlap)
    ms = timed(lambda n: (st.step(n), st.sync()), a.sweeps)
    print(f"  native stepper, RCCL loopback, {'overlap (interior || exchange + edges)' if overlap else 'no overlap (exchange, interior, edges)'}: "
          f"{ms:.4f} ms, halo bytes {st.halo_bytes_per_sweep()}")
    st.close()
