"""One rank of N's j-slab on one GPU through the NATIVE stepper with RCCL in loopback (the rank is
its own neighbour): what the exchange + the edge launches cost against the bare slab sweep.
python profiles/slab_loopback.py [--nj 512] [--dtype f64] [--skew-us 0 100 500 1000 1500 2000 3000]

--skew-us: the one thing loopback cannot show by itself is a neighbour that is LATE.  With a skew the exchange of every
sweep starts that many microseconds late on the communication stream (amt_slab_set_skew_us: a device-side delay in front
of the ncclSend/ncclRecv group), so this rank's halo rows arrive late while its interior rows compute: the sweep time
against the skew shows how much skew the overlap absorbs (about the interior's run time minus exchange and edge rows)
before it shows up one for one -- i.e. how far neighbours may drift apart before the single-buffered halo rows of
profiles/NOTES_r01_r04.md section 9.1 would need a second buffer."""
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
ap.add_argument("--skew-us", type=int, nargs="*", default=[])
ap.add_argument("--transport", choices=("rccl", "ipc"), default="rccl",
                help="ipc: the copy-engine / mailbox transport in loopback (no RCCL kernel; AMT_IPC_PULL=kernel for the one-launch pull)")
ap.add_argument("--beside-rounds", type=int, default=0, help="least rounds of workgroups of the interior launch (amt_march_set_beside; 0 = default)")
ap.add_argument("--beside-reserve", type=int, default=0, help="compute units every round of the interior launch leaves free")
a = ap.parse_args()
pkg = g.load_package()
if a.beside_rounds or a.beside_reserve:
    pkg.load_library().amt_march_set_beside(a.beside_rounds, a.beside_reserve)
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
    st = pkg.patch.NativeSlabStepper(dev, 0, 1, pkg.patch.NativeSlabStepper.comm_unique_id(), loopback=True, overlap=overlap, transport=a.transport)
    ms = timed(lambda n: (st.step(n), st.sync()), a.sweeps)
    print(f"  native stepper, {a.transport.upper()} loopback, {'overlap (interior || exchange + edges)' if overlap else 'no overlap (exchange, interior, edges)'}: "
          f"{ms:.4f} ms, halo bytes {st.halo_bytes_per_sweep()}{', pull by ' + st.pull_mode() if st.pull_mode() else ''}, interior {pkg.load_library().amt_march_last_kernel().decode().split('>')[-1].strip()}")
    st.close()

if a.skew_us:
    print("skew of the neighbours' rows (us) -> ms per sweep, overlap on / off")
    steppers = {ov: pkg.patch.NativeSlabStepper(dev, 0, 1, pkg.patch.NativeSlabStepper.comm_unique_id(), loopback=True, overlap=ov, transport=a.transport)
                for ov in (True, False)}
    for us in a.skew_us:
        row = []
        for ov in (True, False):
            st = steppers[ov]
            st.set_skew_us(us)
            row.append(timed(lambda n: (st.step(n), st.sync()), max(20, a.sweeps // 4)))
        print(f"  {us:6d} us: {row[0]:.4f} ms   {row[1]:.4f} ms   (exposed with overlap: {max(0.0, row[0] - base_overlap) * 1e3:.0f} us)"
              if 'base_overlap' in dir() else f"  {us:6d} us: {row[0]:.4f} ms   {row[1]:.4f} ms")
        if us == 0:
            base_overlap = row[0]
    for st in steppers.values():
        st.close()
