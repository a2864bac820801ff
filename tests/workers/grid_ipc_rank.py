"""One rank of a pi x pj multi-process run on ONE device (tests/test_gpu_33_grid_native.py starts pi * pj of these): the
native stepper amt_grid_* with the IPC halo transport -- HIP pack / unpack of the halo columns, rows in place.  The rank fills
its patch from the generator and poisons every halo row and column it has a neighbour for with NaN; per sweep it gives the fields
that cross a patch boundary new values (seed + sweep, as advance_uv would), poisons the halos again and steps once; it writes the
cells it owns of every output to <dir>/out_<rank>_<name>.npy; the parent holds the unsplit oracle run."""
import argparse
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--grid", type=int, nargs=2, required=True, metavar=("PI", "PJ"))
    ap.add_argument("--dir", required=True)
    ap.add_argument("--dims", type=int, nargs=3, required=True)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--seed", type=int, default=17)
    ap.add_argument("--sweeps", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--specified", action="store_true")
    ap.add_argument("--align", type=int, default=32)
    ap.add_argument("--static-inputs", action="store_true", help="the inputs of sweep 1 for every sweep (see slab_ipc_rank.py)")
    a = ap.parse_args()
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    S, L = pkg.synth, pkg.load_library()
    torch.cuda.set_device(0)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    dims, (pi, pj) = tuple(a.dims), a.grid
    world = pi * pj
    ri, rj = a.rank % pi, a.rank // pi
    pb = S.patch_bounds(S.domain_bounds(*dims), ri, rj, pi, pj, align_elems=a.align)
    cfg = pkg.GridConfig(specified=a.specified)
    uid = (ctypes.c_char * 128)()
    pkg.lib.check(L.amt_comm_rendezvous_file(str(Path(a.dir) / "uid").encode(), 0, a.rank, world, 90.0, uid))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        dev = S.make_patch(pb, cfg, dtype=dtype, seed=a.seed, global_dims=dims, device="cuda:0")
        arr = dev.arrays
        S.poison_halos(dev, S.neighbour_sides(ri, rj, pi, pj))
    torch.cuda.synchronize()
    st = pkg.patch.NativeGridStepper(dev, ri, rj, pi, pj, bytes(uid), stream=stream, overlap=not a.no_overlap, transport="ipc")
    try:
        seen = st.comm_info()
        if a.static_inputs:
            st.step(a.sweeps)
        else:
            for sweep in range(a.sweeps):          # new u, v, t_1 ... every sweep (the stand-in for advance_uv), halos re-poisoned
                if sweep:
                    st.next_substep_inputs(a.seed, sweep)
                st.step(1)
        st.sync()
        own = (slice(pb.jts - pb.jms, pb.jte - pb.jms + 1), Ellipsis, slice(pb.its - pb.ims, pb.ite - pb.ims + 1))
        for n in S.OUTPUTS:
            np.save(Path(a.dir) / f"out_{a.rank}_{n}.npy", arr[n][own].cpu().numpy())
        print(f"rank {a.rank} = patch ({ri},{rj}) of {pi}x{pj}: i {pb.its}..{pb.ite} j {pb.jts}..{pb.jte}, transport {st.transport()}, "
              f"ranks seen {seen[1]}, pull by {st.pull_mode()}, halo bytes {st.halo_bytes_per_sweep()}", flush=True)
    finally:
        st.close()


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    main()
