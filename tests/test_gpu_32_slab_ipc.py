"""Row (e) with more than one REAL rank on a GPU: `world` processes share cuda:0, each owns one j-slab in the native
stepper (amt_slab_*), and the halo rows travel through the IPC transport (hipIpcMemHandles + a shared-memory mailbox +
copy-engine pulls; RCCL refuses two ranks on one device).  Every sweep gets NEW values in the fields that cross a slab
boundary (seed + sweep, the stand-in for advance_uv) and NaN-poisoned halo rows, so only an exchange that delivers every
sweep gives the bits of the UNSPLIT oracle run over the whole domain (what the halo must contain:
advance_mu_t_no_async.cu:121-162, which re-uploads it on every call; module_small_step_em.f90:143-144,241-242).
tests/test_gpu_34_halo_freshness.py proves that these checks turn red on an exchange that stops delivering."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

from multirank import SLAB_WORKER as WORKER, run_slab_ranks, slab_mismatches

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _check_against_the_unsplit_oracle(pkg, oracle, tmp_path, world, dims, dtype, sweeps, specified=False, seed=11):
    bad = slab_mismatches(pkg, oracle, tmp_path, world, dims, dtype, sweeps, specified=specified, seed=seed)
    assert not bad, f"(rank, array) pairs that differ from the unsplit oracle run: {bad}"


@pytest.mark.parametrize("overlap,pull,host_wait", [(True, "kernel", "1"), (True, "kernel", "0"), (False, "kernel", "1"),
                                                    (True, "engine", "1"), (True, "engine", "0")],
                         ids=["host-waited-kernel-pull", "device-waited-fused-kernel", "no-overlap", "host-waited-copy-engine",
                              "device-waited-copy-engine"])
def test_two_processes_on_one_device_at_4096x60x64_per_rank(pkg, oracle, tmp_path, overlap, pull, host_wait):
    """pull: ranks that share a device pull with a kernel by default; "engine" forces the path GPUs of a node take (one
    hipMemcpyAsync per row from the peer mapping).  host_wait: the default schedule posts "rows final" on the domain's stream,
    waits for the neighbours on the HOST and pulls behind a one-round interior; "0" is the device-side wait (a waiting kernel
    enqueued before an interior planned in rounds).  Three sweeps, new inputs and re-poisoned halos before the 2nd and 3rd."""
    dims = (4096, 60, 128)
    outs = run_slab_ranks(tmp_path, 2, dims, sweeps=3, overlap=overlap, extra_env={"AMT_IPC_PULL": pull, "AMT_IPC_HOST_WAIT": host_wait})
    assert all("transport ipc, ranks seen 2" in o and "new every sweep" in o for o in outs), outs
    assert all(("kernel" if pull == "kernel" else "copy engine") in o for o in outs), outs
    _check_against_the_unsplit_oracle(pkg, oracle, tmp_path, 2, dims, "f64", 3)


@pytest.mark.parametrize("host_wait", ["1", "0"], ids=["host-waited", "device-waited"])
def test_three_processes_uneven_rows_specified_boundaries_fp32(pkg, oracle, tmp_path, host_wait):
    """The middle rank has both neighbours; 61 rows over 3 ranks; `specified` clips the outermost rows."""
    dims = (300, 24, 61)
    run_slab_ranks(tmp_path, 3, dims, dtype="f32", sweeps=3, specified=True, extra_env={"AMT_IPC_HOST_WAIT": host_wait})
    _check_against_the_unsplit_oracle(pkg, oracle, tmp_path, 3, dims, "f32", 3, specified=True)


@pytest.mark.parametrize("pull", ["kernel", "engine"])
def test_rows_that_are_not_a_multiple_of_16_bytes(pkg, oracle, tmp_path, pull):
    dims = (515, 33, 40)                   # 517-element fp32 rows: the pull kernels' unaligned and tail paths
    run_slab_ranks(tmp_path, 2, dims, dtype="f32", sweeps=3, extra_env={"AMT_IPC_PULL": pull})
    _check_against_the_unsplit_oracle(pkg, oracle, tmp_path, 2, dims, "f32", 3)


def test_a_missing_neighbour_ends_with_a_diagnosis_not_a_hang(pkg, tmp_path):
    """Rank 1 of 2 never starts: rank 0's set-up gives up after AMT_IPC_TIMEOUT_S with AMT_ERR_COMM (non-zero exit, text)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(AMT_RENDEZVOUS_NONCE=f"ipc-{tmp_path.name}", AMT_SLAB_TRANSPORT="ipc", AMT_IPC_TIMEOUT_S="5")
    # give rank 0 its id without a peer: a pre-made rendezvous is not possible, so start rank 1 and let it die before the set-up
    code = ("import sys, ctypes; sys.path.insert(0, %r); import __graft_entry__ as g; pkg = g.load_package(); L = pkg.load_library();"
            "uid = (ctypes.c_char * 128)(); pkg.lib.check(L.amt_comm_rendezvous_file(%r.encode(), 0, 1, 2, 60.0, uid))"
            % (str(ROOT), str(tmp_path / "uid")))
    quitter = subprocess.Popen([sys.executable, "-c", code], env=env)
    r = subprocess.run([sys.executable, str(WORKER), "--rank", "0", "--world", "2", "--dir", str(tmp_path), "--dims", "64", "10", "16"],
                       env=env, capture_output=True, text=True, timeout=300)
    quitter.wait(timeout=60)
    assert r.returncode != 0
    assert "did not publish" in r.stdout + r.stderr or "ranks attached" in r.stdout + r.stderr, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("host_wait,pull", [("1", "kernel"), ("0", "kernel"), ("1", "engine"), ("0", "engine")],
                         ids=["host-waited", "device-waited", "host-waited-copy-engine", "device-waited-copy-engine"])
def test_three_ranks_drifting_apart_for_150_sweeps(pkg, oracle, tmp_path, host_wait, pull):
    """Stress of the mailbox protocol: three ranks, 150 single-sweep calls each with a random host sleep of up to 400 us in front
    of every call, so that neighbours are early and late in turn while staging buffers are refreshed and pulled.  Every sweep
    sends DIFFERENT rows (inputs of seed + sweep) into NaN-poisoned halos: a pull that came before the neighbour's refresh, a
    refresh that came before the neighbour's pull, or a sweep without a delivery leaves a wrong or NaN row in t, mu, ww, which
    accumulate -- the result after 150 sweeps is the unsplit oracle run's, bit for bit, only if all 150 exchanges delivered
    that sweep's rows."""
    dims = (96, 12, 30)
    run_slab_ranks(tmp_path, 3, dims, sweeps=150, jitter_us=400, extra_env={"AMT_IPC_HOST_WAIT": host_wait, "AMT_IPC_PULL": pull})
    _check_against_the_unsplit_oracle(pkg, oracle, tmp_path, 3, dims, "f64", 150)
