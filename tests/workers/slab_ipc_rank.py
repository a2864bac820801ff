"""One rank of a multi-process j-slab run on ONE device (tests/test_gpu_32_slab_ipc.py and test_gpu_34_halo_freshness.py start
`world` of these): the native stepper amt_slab_* with the IPC halo transport.  The rank fills its slab from the generator and
poisons its halo rows with NaN; then, per sweep, it gives the fields that cross a slab boundary NEW values (seed + sweep: the
stand-in for advance_uv rewriting u, v before every call, module_small_step_em.f90:143-146,241-245), poisons the halo rows
again and steps once -- so only an exchange that delivers EVERY sweep gives the unsplit oracle run's bits.  It writes the rows
it owns of every output to <dir>/out_<rank>_<name>.npy for the parent, which holds that oracle run.  Host-side rendezvous: the
library's own file rendezvous (amt_comm_rendezvous_file), as a Fortran or C host without MPI would do it.
--static-inputs keeps the inputs of sweep 1 for every sweep (what a static benchmark does: an exchange that works once then
satisfies all later sweeps -- the freshness tests use it to show exactly that)."""
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
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--dims", type=int, nargs=3, required=True)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--sweeps", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--specified", action="store_true")
    ap.add_argument("--transport", default="ipc")
    ap.add_argument("--static-inputs", action="store_true")
    ap.add_argument("--jitter-us", type=int, default=0,
                    help="stress: a random host sleep of up to this many microseconds in front of each sweep")
    a = ap.parse_args()
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    S, L = pkg.synth, pkg.load_library()
    torch.cuda.set_device(0)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    dims = tuple(a.dims)
    gb = S.domain_bounds(*dims, aligned=True)
    sb = S.slab_bounds(gb, a.rank, a.world)
    cfg = pkg.GridConfig(specified=a.specified)
    uid = (ctypes.c_char * 128)()
    pkg.lib.check(L.amt_comm_rendezvous_file(str(Path(a.dir) / "uid").encode(), 0, a.rank, a.world, 90.0, uid))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        dev = S.make_patch(sb, cfg, dtype=dtype, seed=a.seed, global_dims=dims, device="cuda:0")
        S.poison_halos(dev, S.neighbour_sides(0, a.rank, 1, a.world))
    torch.cuda.synchronize()
    st = pkg.patch.NativeSlabStepper(dev, a.rank, a.world, bytes(uid), stream=stream, overlap=not a.no_overlap,
                                     transport=a.transport)
    try:
        assert st.transport() == a.transport, st.transport()
        seen = st.comm_info()
        import random
        import time
        rng = random.Random(1000 + a.rank)
        if a.static_inputs and not a.jitter_us:
            st.step(a.sweeps)
        else:
            for sweep in range(a.sweeps):
                if a.jitter_us:                     # ranks drift apart by up to the jitter every sweep, in both directions
                    time.sleep(rng.random() * a.jitter_us * 1e-6)
                if sweep and not a.static_inputs:
                    st.next_substep_inputs(a.seed, sweep)      # on the domain's stream: new u, v, t_1 ...; NaN in the halo rows
                st.step(1)
        st.sync()                                   # raises AmtError(ERR_COMM) if a device-side wait gave up
        for n in S.OUTPUTS:
            np.save(Path(a.dir) / f"out_{a.rank}_{n}.npy", dev.arrays[n][1:-1].cpu().numpy())
        print(f"rank {a.rank}/{a.world}: rows {sb.jts}..{sb.jte}, transport {st.transport()}, ranks seen {seen[1]}, pull by {st.pull_mode()}, "
              f"kernel {L.amt_march_last_kernel().decode()}, inputs {'static' if a.static_inputs else 'new every sweep'}", flush=True)
    finally:
        st.close()


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    main()
