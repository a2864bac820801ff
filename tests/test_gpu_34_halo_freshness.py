"""Can the multi-rank tests tell a working halo exchange from one that works ONCE?  

What crosses a patch boundary are pure inputs of advance_mu_t (module_small_step_em.f90:143-146, 241-245); in WRF advance_uv
rewrites them before every call and the reference re-uploads them on every call (advance_mu_t_no_async.cu:245-306).  The tests
of rows (e) and (f4) therefore give every sweep its own values of those fields and re-poison the halos (tests/multirank.py).
This file proves that this makes them sensitive: each fault of AMT_TEST_FAULT (csrc/amt_internal.h) -- a staging buffer that is
not refreshed, a pull that does not happen, an RCCL group that is not issued, halo columns that are not gathered / scattered,
each on ONE exchange after the first -- must turn the comparison with the oracle RED, in every schedule (no overlap, host-waited,
device-waited), with both pulls (fused kernel, copy engine), in loopback and between real processes; and it shows the blind spot
itself: with static inputs the same faults are invisible."""
import os

import numpy as np
import pytest

from conftest import bits_equal
from multirank import (grid_mismatches, loopback_halos_by_hand, oracle_sweeps, run_grid_ranks, run_slab_ranks, slab_mismatches)

pytestmark = pytest.mark.gpu
SWEEPS = 4


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch


class _Fault:
    """AMT_TEST_FAULT for the duration of a with-block (the library reads it per exchange)."""

    def __init__(self, spec):
        self.spec = spec

    def __enter__(self):
        self.old = os.environ.pop("AMT_TEST_FAULT", None)
        if self.spec:
            os.environ["AMT_TEST_FAULT"] = self.spec

    def __exit__(self, *exc):
        os.environ.pop("AMT_TEST_FAULT", None)
        if self.old is not None:
            os.environ["AMT_TEST_FAULT"] = self.old


def _loopback_differs(pkg, oracle, torch_mod, *, grid, transport, overlap, fault, static=False, dtype=np.float64, seed=57):
    """One rank as its own neighbour (rows; rows and columns for grid=True), SWEEPS sweeps with new inputs and re-poisoned
    halos before each but the first (static: the same inputs throughout, halos poisoned once), under `fault`.  Returns the
    outputs whose owned cells differ from the oracle run with the halos copied by hand."""
    S = pkg.synth
    gdims = (190, 14, 45)
    if grid:
        pb = S.patch_bounds(S.domain_bounds(*gdims), 1, 1, 3, 3, align_elems=32)
    else:
        pb = S.slab_bounds(S.domain_bounds(*gdims, aligned=True), 1, 3)
    dev = S.make_patch(pb, pkg.GridConfig(), dtype=dtype, seed=seed, global_dims=gdims, device="cuda:0")
    want = dev.to_host()
    S.poison_halos(dev, 15 if grid else S.SIDE_BELOW | S.SIDE_ABOVE)
    torch_mod.cuda.synchronize()
    Stepper = pkg.patch.NativeGridStepper if grid else pkg.patch.NativeSlabStepper
    args = (dev, 0, 0, 1, 1) if grid else (dev, 0, 1)
    with _Fault(fault):
        st = Stepper(*args, Stepper.comm_unique_id(), loopback=True, overlap=overlap, transport=transport)
        try:
            for sweep in range(SWEEPS):
                if sweep and not static:
                    st.next_substep_inputs(seed, sweep)
                st.step(1)
            st.sync()
        finally:
            st.close()
    oracle_sweeps(pkg, oracle, want, seed, SWEEPS, refresh=not static,
                  before_each=lambda p: loopback_halos_by_hand(pkg, p, columns=grid))
    got = dev.to_host()
    own = (slice(pb.jts - pb.jms, pb.jte - pb.jms + 1), Ellipsis, slice(pb.its - pb.ims, pb.ite - pb.ims + 1))
    return [n for n in S.OUTPUTS if not bits_equal(got.arrays[n][own], want.arrays[n][own])]


SLAB_FAULTS = [("rccl", "skip_group@2"), ("rccl", "skip_group@4"), ("ipc", "skip_stage@2"), ("ipc", "skip_stage@4"), ("ipc", "skip_pull@3")]


@pytest.mark.parametrize("overlap", [True, False], ids=["overlap", "no-overlap"])
@pytest.mark.parametrize("transport,fault", SLAB_FAULTS, ids=[f"{t}-{f}" for t, f in SLAB_FAULTS])
def test_loopback_slab_one_faulty_exchange_after_the_first_turns_the_check_red(pkg, oracle, torch_mod, transport, fault, overlap):
    assert _loopback_differs(pkg, oracle, torch_mod, grid=False, transport=transport, overlap=overlap, fault=None) == []
    bad = _loopback_differs(pkg, oracle, torch_mod, grid=False, transport=transport, overlap=overlap, fault=fault)
    assert bad, f"{fault} went unnoticed"


GRID_FAULTS = [("rccl", "skip_group@3"), ("ipc", "skip_stage@2"), ("ipc", "skip_pull@4"), ("rccl", "skip_pack@2"), ("ipc", "skip_pack@3"),
               ("rccl", "skip_unpack@2"), ("ipc", "skip_unpack@4")]


@pytest.mark.parametrize("overlap", [True, False], ids=["overlap", "no-overlap"])
@pytest.mark.parametrize("transport,fault", GRID_FAULTS, ids=[f"{t}-{f}" for t, f in GRID_FAULTS])
def test_loopback_grid_one_faulty_exchange_after_the_first_turns_the_check_red(pkg, oracle, torch_mod, transport, fault, overlap):
    assert _loopback_differs(pkg, oracle, torch_mod, grid=True, transport=transport, overlap=overlap, fault=None) == []
    bad = _loopback_differs(pkg, oracle, torch_mod, grid=True, transport=transport, overlap=overlap, fault=fault)
    assert bad, f"{fault} went unnoticed"


@pytest.mark.parametrize("transport,fault", [("rccl", "skip_group@2"), ("ipc", "skip_stage@2"), ("ipc", "skip_pull@3")])
def test_the_blind_spot_with_static_inputs_the_same_faults_are_invisible(pkg, oracle, torch_mod, transport, fault):
    """Why the inputs must change: halos poisoned once, inputs constant -- after the first exchange the halo holds the right
    values for ever, and an exchange that is skipped, or that delivers the previous sweep's rows, gives the same bits."""
    assert _loopback_differs(pkg, oracle, torch_mod, grid=False, transport=transport, overlap=True, fault=fault, static=True) == []


@pytest.mark.parametrize("host_wait,pull", [("1", "kernel"), ("0", "kernel"), ("1", "engine"), ("0", "engine")],
                         ids=["host-waited-kernel-pull", "device-waited-fused-kernel", "host-waited-copy-engine", "device-waited-copy-engine"])
@pytest.mark.parametrize("fault", ["skip_stage@2", "skip_stage@4", "skip_pull@3"])
def test_two_processes_a_stale_or_missing_delivery_turns_the_check_red(pkg, oracle, tmp_path, fault, host_wait, pull):
    """Two real processes on cuda:0, every schedule and pull of the IPC transport: a staging buffer that is not refreshed on
    exchange n (the neighbour then pulls the previous sweep's rows) or rows that are not pulled on exchange n (the halo
    stays NaN) must show against the unsplit oracle run."""
    dims = (200, 16, 24)
    run_slab_ranks(tmp_path, 2, dims, sweeps=SWEEPS, fault=fault, extra_env={"AMT_IPC_PULL": pull, "AMT_IPC_HOST_WAIT": host_wait})
    bad = slab_mismatches(pkg, oracle, tmp_path, 2, dims, "f64", SWEEPS)
    assert bad, f"{fault} went unnoticed"
    assert {r for r, _ in bad} == {0, 1}, bad               # both ranks receive from the other: both must be wrong


def test_two_processes_healthy_and_the_blind_spot(pkg, oracle, tmp_path):
    dims = (200, 16, 24)
    (tmp_path / "ok").mkdir()
    run_slab_ranks(tmp_path / "ok", 2, dims, sweeps=SWEEPS)
    assert slab_mismatches(pkg, oracle, tmp_path / "ok", 2, dims, "f64", SWEEPS) == []
    (tmp_path / "static").mkdir()
    run_slab_ranks(tmp_path / "static", 2, dims, sweeps=SWEEPS, static_inputs=True, fault="skip_stage@2")
    assert slab_mismatches(pkg, oracle, tmp_path / "static", 2, dims, "f64", SWEEPS, static_inputs=True) == [], \
        "with static inputs a stale staging buffer is expected to be invisible"


@pytest.mark.parametrize("host_wait", ["1", "0"], ids=["host-waited", "device-waited"])
@pytest.mark.parametrize("fault", ["skip_pack@2", "skip_unpack@3", "skip_stage@3"])
def test_2x2_processes_stale_columns_or_rows_turn_the_check_red(pkg, oracle, tmp_path, fault, host_wait):
    dims = (150, 12, 40)
    run_grid_ranks(tmp_path, 2, 2, dims, sweeps=SWEEPS, fault=fault, host_wait=host_wait)
    bad = grid_mismatches(pkg, oracle, tmp_path, 2, 2, dims, "f64", SWEEPS, False, 32)
    assert bad, f"{fault} went unnoticed"
    assert {r for r, _ in bad} == {0, 1, 2, 3}, bad
