"""One patch of a pi x pj decomposition on one GPU through the NATIVE grid stepper (amt_grid_*) in loopback (the rank is its own
neighbour on all four sides): what the column pack / unpack, the exchange and the boundary tiles cost against the bare patch sweep.
python profiles/grid_loopback.py [--ni 2048 --nj 2048] [--transport ipc|rccl]"""
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
ap.add_argument("--sweeps", type=int, default=50)
ap.add_argument("--transport", choices=("rccl", "ipc"), default="ipc")
a = ap.parse_args()
pkg = g.load_package()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
gdims = (3 * a.ni, a.nk, 3 * a.nj)
pb = S.patch_bounds(S.domain_bounds(*gdims), 1, 1, 3, 3, align_elems=32)      # the middle patch of 3 x 3


def timed(fn, n):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


dev = S.make_patch(pb, pkg.GridConfig(), dtype=dtype, seed=1, global_dims=gdims, device="cuda:0")
bare = pkg.bind_device_call(*dev.args())
ms_bare = timed(lambda n: [bare() for _ in range(n)], a.sweeps)
cells = a.ni * a.nk * a.nj
print(f"{a.ni}x{a.nk}x{a.nj} {a.dtype} patch, bare sweep (one launch): {ms_bare:.4f} ms = {cells / ms_bare / 1e6:.2f} Gcells/s")
for overlap in (True, False):
    st = pkg.patch.NativeGridStepper(dev, 0, 0, 1, 1, pkg.patch.NativeGridStepper.comm_unique_id(), loopback=True, overlap=overlap,
                                     transport=a.transport)
    ms = timed(lambda n: (st.step(n), st.sync()), a.sweeps)
    print(f"  native grid stepper, {a.transport.upper()} loopback, {'overlap' if overlap else 'no overlap'}: {ms:.4f} ms (+{100 * (ms / ms_bare - 1):.1f} %), "
          f"halo bytes {st.halo_bytes_per_sweep()}, pull by {st.pull_mode() or '-'}")
    st.close()
