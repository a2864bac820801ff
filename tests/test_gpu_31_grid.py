"""i x j decomposition on a real GPU through patch.GridStepper: four processes (2 x 2 patches) share cuda:0, NaN-poisoned halo
cells.  Two paths behind the one class: the torch.distributed bring-up path (halos staged through the host over gloo) and the
native one (GridStepper(native="ipc") forwards to the C++ runtime amt_grid_*: HIP pack / unpack kernels, IPC transport).
Every sweep has its own values of the exchanged fields and freshly poisoned halos (tests/multirank.py says why)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, pi, pj, sweeps, out_dir, native=None):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        pkg = g.load_package()
        S = pkg.synth
        torch.cuda.set_device(0)
        ri, rj = rank % pi, rank // pi
        pb = S.patch_bounds(S.domain_bounds(*shape), ri, rj, pi, pj, align_elems=32)
        dev = S.make_patch(pb, pkg.GridConfig(specified=True), seed=17, global_dims=shape, device="cuda:0")
        a, sides = dev.arrays, S.neighbour_sides(ri, rj, pi, pj)
        S.poison_halos(dev, sides)
        if native:
            # GridStepper as a thin caller of the C++ runtime (amt_grid_*): the communicator id travels over the host group
            uid = [pkg.patch.NativeGridStepper.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            st = pkg.patch.GridStepper(dev, ri, rj, pi, pj, pkg.advance_mu_t, native=native, unique_id=uid[0])
        else:
            st = pkg.patch.GridStepper(dev, ri, rj, pi, pj, pkg.advance_mu_t, stage_through_host=True)
        for sweep in range(sweeps):
            if sweep:                # new u, v, t_1 ... before every sweep but the first (the stand-in for advance_uv), halos NaN again
                if native:
                    st._native.next_substep_inputs(17, sweep)
                else:
                    S.refresh_exchanged_inputs(dev, 17, sweep)
                    S.poison_halos(dev, sides)
            st.step()
        if native:
            st._native.sync()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), bounds=np.array(pb.as_tuple()),
                 **{n: a[n].cpu().numpy() for n in S.OUTPUTS})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("native", [None, "ipc"], ids=["torch-gloo-host-staged", "native-amt_grid-ipc"])
def test_2x2_patches_on_one_gpu_match_the_oracle(tmp_path, pkg, oracle, native, monkeypatch):
    shape, pi, pj, sweeps = (150, 12, 40), 2, 2, 3
    if native:
        monkeypatch.setenv("AMT_SLAB_TRANSPORT", "ipc")           # rank 0's id then needs no RCCL
        monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_worker, args=(pi * pj, _free_port(), shape, pi, pj, sweeps, str(tmp_path), native), nprocs=pi * pj, join=True)
    S = pkg.synth
    full = S.make_patch(S.domain_bounds(*shape), pkg.GridConfig(specified=True), seed=17)
    for sweep in range(sweeps):
        if sweep:
            S.refresh_exchanged_inputs(full, 17, sweep)
        oracle.advance_mu_t(*full.args())
    for rank in range(pi * pj):
        r = np.load(tmp_path / f"rank{rank}.npz")
        b = S.Bounds(*[int(x) for x in r["bounds"]])
        for n in S.OUTPUTS:
            mine = r[n][b.jts - b.jms: b.jte - b.jms + 1, ..., b.its - b.ims: b.ite - b.ims + 1]
            want = full.arrays[n][b.jts: b.jte + 1, ..., b.its: b.ite + 1]
            assert np.array_equal(mine.view(np.uint8), want.view(np.uint8)), (rank, n)
