"""N > 1 path on CPU: world_size 2 and 3 over gloo.  The j-slab decomposition + one-row
input-halo exchange (wrf-model-cuda-sample_amd/patch.py, the code bench.py runs over RCCL)
must reproduce the unsplit domain bit for bit.  The compute callable is injected: here it
is the oracle (tests only); the product passes the HIP entry point."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, flags, sweeps, out_dir, static=False):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        pkg, oracle = g.load_package(), g.load_oracle()
        S = pkg.synth
        ni, nk, nj = shape
        gb = S.domain_bounds(ni, nk, nj)
        sb = S.slab_bounds(gb, rank, world)
        host = S.make_patch(sb, pkg.GridConfig(**flags), seed=77, global_dims=shape)
        arrays = {k: torch.from_numpy(v) for k, v in host.arrays.items()}
        patch = S.Patch(sb, host.config, arrays, host.rdx, host.rdy, host.dts, host.epssm, shape)

        def compute(*args):
            args = [a.numpy() if isinstance(a, torch.Tensor) else a for a in args]
            oracle.advance_mu_t(*args)

        stepper = pkg.patch.SlabStepper(patch, rank, world, compute)
        for sweep in range(sweeps):
            if sweep and not static:                    # new u, v, t_1 ... every sweep: the stand-in for advance_uv
                S.refresh_exchanged_inputs(patch, 77, sweep)
            # poison the halo rows: only an exchange that delivers THIS sweep's rows can make the result right
            if not (sweep and static):
                S.poison_halos(patch, S.neighbour_sides(0, rank, 1, world))
            stepper.step()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), jms=sb.jms, jts=sb.jts, jte=sb.jte,
                 halo_bytes=stepper.halo_bytes_per_sweep(),
                 **{n: arrays[n].numpy() for n in S.OUTPUTS})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,flags", [
    (2, (33, 7, 10), dict()),
    (2, (20, 5, 9), dict(specified=True)),
    (3, (18, 4, 7), dict(nested=True)),      # middle slab has two neighbours; a 2-row slab
    (3, (12, 3, 3), dict()),                 # one-row slabs
    (8, (24, 5, 61), dict(specified=True)),  # the target N, uneven rows (7 and 8 per rank), clipped outermost rows
])
def test_slabs_reproduce_the_unsplit_domain(tmp_path, world, shape, flags):
    sweeps = 3
    mp.spawn(_worker, args=(world, _free_port(), shape, flags, sweeps, str(tmp_path)), nprocs=world, join=True)

    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as g
    pkg, oracle = g.load_package(), g.load_oracle()
    S = pkg.synth
    full = S.make_patch(S.domain_bounds(*shape), pkg.GridConfig(**flags), seed=77)
    for sweep in range(sweeps):
        if sweep:
            S.refresh_exchanged_inputs(full, 77, sweep)
        oracle.advance_mu_t(*full.args())
    total_halo = 0
    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        jms, jts, jte = int(r["jms"]), int(r["jts"]), int(r["jte"])
        total_halo += int(r["halo_bytes"])
        for n in S.OUTPUTS:
            mine = r[n][jts - jms: jte - jms + 1]
            want = full.arrays[n][jts: jte + 1]          # global jms = 0
            assert np.array_equal(mine.view(np.uint8), want.view(np.uint8)), (rank, n)
    ni, nk, _ = shape
    idim, kdim = ni + 2, nk + 1
    per_interface = 8 * (idim * kdim * 4 + idim * 2)     # W*idim*kdim*4 + W*idim*2
    assert total_halo == 2 * (world - 1) * per_interface  # counted on both sides of each interface


def test_slab_bounds_partition_the_rows(pkg):
    S = pkg.synth
    g = S.domain_bounds(8, 3, 37)
    rows = []
    for r in range(5):
        b = S.slab_bounds(g, r, 5)
        assert (b.jms, b.jme) == (b.jts - 1, b.jte + 1)
        assert (b.ids, b.ide, b.jds, b.jde) == (g.ids, g.ide, g.jds, g.jde)
        rows += list(range(b.jts, b.jte + 1))
    assert rows == list(range(1, 38))
    with pytest.raises(ValueError):
        S.slab_bounds(S.domain_bounds(8, 3, 2), 0, 3)


def _grid_worker(rank, world, port, shape, flags, pi, pj, sweeps, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        pkg, oracle = g.load_package(), g.load_oracle()
        S = pkg.synth
        ri, rj = rank % pi, rank // pi
        pb = S.patch_bounds(S.domain_bounds(*shape), ri, rj, pi, pj)
        host = S.make_patch(pb, pkg.GridConfig(**flags), seed=91, global_dims=shape)
        arrays = {k: torch.from_numpy(v) for k, v in host.arrays.items()}
        patch = S.Patch(pb, host.config, arrays, host.rdx, host.rdy, host.dts, host.epssm, shape)

        def compute(*args):
            oracle.advance_mu_t(*[a.numpy() if isinstance(a, torch.Tensor) else a for a in args])

        st = pkg.patch.GridStepper(patch, ri, rj, pi, pj, compute)
        for sweep in range(sweeps):
            if sweep:
                S.refresh_exchanged_inputs(patch, 91, sweep)
            S.poison_halos(patch, S.neighbour_sides(ri, rj, pi, pj))
            st.step()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), bounds=np.array(pb.as_tuple()),
                 **{n: arrays[n].numpy() for n in S.OUTPUTS})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pi,pj,shape,flags", [
    (2, 1, (21, 5, 6), dict()),
    (2, 2, (17, 4, 9), dict(specified=True)),
    (3, 2, (20, 3, 7), dict(nested=True, periodic_x=True)),
    (4, 2, (29, 3, 11), dict(specified=True)),            # eight ranks, uneven in both directions
])
def test_2d_patches_reproduce_the_unsplit_domain(tmp_path, pi, pj, shape, flags):
    """i x j decomposition: packed column halos + row halos, NaN-poisoned."""
    sweeps, world = 3, pi * pj
    mp.spawn(_grid_worker, args=(world, _free_port(), shape, flags, pi, pj, sweeps, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as g
    pkg, oracle = g.load_package(), g.load_oracle()
    S = pkg.synth
    full = S.make_patch(S.domain_bounds(*shape), pkg.GridConfig(**flags), seed=91)
    for sweep in range(sweeps):
        if sweep:
            S.refresh_exchanged_inputs(full, 91, sweep)
        oracle.advance_mu_t(*full.args())
    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        b = S.Bounds(*[int(x) for x in r["bounds"]])
        for n in S.OUTPUTS:
            mine = r[n][b.jts - b.jms: b.jte - b.jms + 1, ..., b.its - b.ims: b.ite - b.ims + 1]
            want = full.arrays[n][b.jts: b.jte + 1, ..., b.its: b.ite + 1]        # global ims = jms = 0
            assert np.array_equal(mine.view(np.uint8), want.view(np.uint8)), (rank, n)


def _slab_run_differs(tmp_path, world, shape, sweeps, *, fault=None, static=False):
    """Ranks whose owned rows differ from the unsplit oracle run after `sweeps` sweeps under AMT_TEST_FAULT=fault."""
    old = os.environ.pop("AMT_TEST_FAULT", None)
    if fault:
        os.environ["AMT_TEST_FAULT"] = fault            # inherited by the spawned ranks
    try:
        mp.spawn(_worker, args=(world, _free_port(), shape, {}, sweeps, str(tmp_path), static), nprocs=world, join=True)
    finally:
        os.environ.pop("AMT_TEST_FAULT", None)
        if old is not None:
            os.environ["AMT_TEST_FAULT"] = old
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as g
    pkg, oracle = g.load_package(), g.load_oracle()
    S = pkg.synth
    full = S.make_patch(S.domain_bounds(*shape), pkg.GridConfig(), seed=77)
    for sweep in range(sweeps):
        if sweep and not static:
            S.refresh_exchanged_inputs(full, 77, sweep)
        oracle.advance_mu_t(*full.args())
    bad = set()
    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        jms, jts, jte = int(r["jms"]), int(r["jts"]), int(r["jte"])
        for n in S.OUTPUTS:
            if not np.array_equal(r[n][jts - jms: jte - jms + 1].view(np.uint8), full.arrays[n][jts: jte + 1].view(np.uint8)):
                bad.add(rank)
    return bad


def test_an_exchange_that_stops_delivering_turns_the_check_red(tmp_path):
    """The sensitivity the per-sweep refresh buys: AMT_TEST_FAULT=skip_exchange@n makes every rank skip
    its n-th exchange.  With inputs that change every sweep and re-poisoned halos all three ranks come out wrong; with static
    inputs (halos poisoned once) the same fault is INVISIBLE -- the blind spot the old tests had."""
    shape, sweeps = (18, 4, 9), 4
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir(); (tmp_path / "c").mkdir()
    assert _slab_run_differs(tmp_path / "a", 3, shape, sweeps) == set()
    assert _slab_run_differs(tmp_path / "b", 3, shape, sweeps, fault="skip_exchange@3") == {0, 1, 2}
    assert _slab_run_differs(tmp_path / "c", 3, shape, sweeps, fault="skip_exchange@3", static=True) == set()
