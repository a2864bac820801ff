"""amt_comm_rendezvous_file: the launch nonce (a file left by an earlier launch on
the same path must never be taken for this launch's).  Ranks other than 0 never touch RCCL, so this
runs without a GPU: the "rank 0" side is written by hand in the documented format
(8-byte magic, 8-byte nonce, the id)."""
import ctypes
import struct
import threading
import time

MAGIC = b"AMTUID02"


def _publish(path, nonce, ident):
    path.write_bytes(MAGIC + struct.pack("<Q", nonce) + ident)


def test_other_launchs_file_is_refused_however_fresh(pkg, tmp_path):
    from wrf_model_cuda_sample_amd import lib
    L = pkg.load_library()
    out = (ctypes.c_char * 128)()
    path = tmp_path / "uid"
    _publish(path, 1234, bytes(range(128)))                     # written just now, by "another launch"
    rc = L.amt_comm_rendezvous_file(str(path).encode(), 999, 1, 2, 0.3, out)
    assert rc == lib.ERR_COMM
    assert b"another launch" in L.amt_last_error()
    assert not (tmp_path / "uid.ack.1").exists()


def test_stale_file_is_skipped_until_this_launchs_arrives(pkg, tmp_path):
    from wrf_model_cuda_sample_amd import lib
    L = pkg.load_library()
    out = (ctypes.c_char * 128)()
    path = tmp_path / "uid"
    ident = bytes((7 * i) % 251 for i in range(128))
    _publish(path, 1, b"\x55" * 128)                            # stale

    def rank0_later():
        time.sleep(0.3)
        tmp = tmp_path / "uid.tmp"
        tmp.write_bytes(MAGIC + struct.pack("<Q", 42) + ident)
        tmp.rename(path)
    t = threading.Thread(target=rank0_later)
    t.start()
    lib.check(L.amt_comm_rendezvous_file(str(path).encode(), 42, 3, 4, 10.0, out))
    t.join()
    assert bytes(out) == ident
    assert (tmp_path / "uid.ack.3").read_bytes() == struct.pack("<Q", 42)


def test_launch_nonce_is_stable_in_a_process_and_follows_the_environment(pkg, monkeypatch):
    L = pkg.load_library()
    monkeypatch.delenv("AMT_RENDEZVOUS_NONCE", raising=False)
    a, b = L.amt_comm_launch_nonce(), L.amt_comm_launch_nonce()
    assert a == b and a != 0
    monkeypatch.setenv("AMT_RENDEZVOUS_NONCE", "launch-1")
    c = L.amt_comm_launch_nonce()
    monkeypatch.setenv("AMT_RENDEZVOUS_NONCE", "launch-2")
    d = L.amt_comm_launch_nonce()
    assert len({a, c, d}) == 3


def test_bad_arguments(pkg, tmp_path):
    from wrf_model_cuda_sample_amd import lib
    L = pkg.load_library()
    out = (ctypes.c_char * 128)()
    p = str(tmp_path / "x").encode()
    assert L.amt_comm_rendezvous_file(p, 1, 2, 2, 0.1, out) == lib.ERR_INVALID_ARG      # rank outside the world
    assert L.amt_comm_rendezvous_file(p, 1, -1, 2, 0.1, out) == lib.ERR_INVALID_ARG
    assert L.amt_comm_rendezvous_file(b"", 1, 1, 2, 0.1, out) == lib.ERR_INVALID_ARG
    assert L.amt_comm_rendezvous_file(p, 1, 1, 2, 0.1, None) == lib.ERR_INVALID_ARG


def test_ranks_of_one_scheduler_job_agree_without_a_common_parent(pkg):
    """per-node srun / orted daemons give the ranks of one job DIFFERENT parents; the job id the
    scheduler exports must then decide the nonce, not the parent process.  Two child processes started through
    two different intermediate shells (different ppid) with the same SLURM_JOB_ID compute the same value; with
    different job ids, or with neither job id nor a common parent, different ones."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; "
            "print(g.load_package().load_library().amt_comm_launch_nonce())" % str(root))

    def via_own_shell(extra):
        env = {k: v for k, v in os.environ.items()
               if k not in ("AMT_RENDEZVOUS_NONCE", "SLURM_JOB_ID", "SLURM_STEP_ID", "PMI_JOBID", "PMIX_NAMESPACE",
                            "TORCHELASTIC_RUN_ID", "LSB_JOBID", "PBS_JOBID", "PMI_ID_JOB", "OMPI_MCA_ess_base_jobid")}
        env.update(extra)
        # `sh -c '...; true'` keeps the shell alive as the python process's parent: one parent per rank
        r = subprocess.run(["sh", "-c", f"{sys.executable} -c \"{code}\"; true"], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-500:]
        return int(r.stdout.strip().splitlines()[-1])

    job = {"SLURM_JOB_ID": "4711", "SLURM_STEP_ID": "0", "MASTER_PORT": "29500"}
    a, b = via_own_shell(job), via_own_shell(job)
    assert a == b and a != 0
    assert via_own_shell(dict(job, SLURM_JOB_ID="4712")) != a
    c, d = via_own_shell({"MASTER_PORT": "29500"}), via_own_shell({"MASTER_PORT": "29500"})
    assert c != d                                  # hand-started ranks: no common parent, no job id -> need AMT_RENDEZVOUS_NONCE
    e, f = via_own_shell({"AMT_RENDEZVOUS_NONCE": "x"}), via_own_shell({"AMT_RENDEZVOUS_NONCE": "x"})
    assert e == f


def test_two_launches_inside_one_scheduler_job_do_not_share_a_nonce(pkg):
    """with a scheduler job id set, two torchrun launches inside the one allocation (or an elastic
    restart on the same port) must still get different nonces -- otherwise ranks >= 1 of the second launch accept the
    rendezvous file a crashed first launch left.  The local launcher (TORCHELASTIC_RUN_ID set) contributes its run id,
    its restart count and itself as the parent process; ranks of ONE such launch (one parent) still agree."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; "
            "print(g.load_package().load_library().amt_comm_launch_nonce())" % str(root))
    base = {k: v for k, v in os.environ.items()
            if k not in ("AMT_RENDEZVOUS_NONCE", "SLURM_JOB_ID", "SLURM_STEP_ID", "PMI_JOBID", "PMIX_NAMESPACE",
                         "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "LSB_JOBID", "PBS_JOBID", "PMI_ID_JOB",
                         "OMPI_MCA_ess_base_jobid")}

    def launch(extra, ranks=1):
        """One 'launcher' shell that starts `ranks` rank processes: they share the parent."""
        env = dict(base, **extra)
        cmd = "; ".join([f"{sys.executable} -c \"{code}\""] * ranks) + "; true"
        r = subprocess.run(["sh", "-c", cmd], env=env, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr[-500:]
        return [int(x) for x in r.stdout.split()]

    job = {"SLURM_JOB_ID": "4711", "MASTER_PORT": "29500", "TORCHELASTIC_RUN_ID": "none", "TORCHELASTIC_RESTART_COUNT": "0"}
    first = launch(job, ranks=2)
    assert first[0] == first[1] != 0                                     # the ranks of one launch agree
    assert launch(job)[0] != first[0]                                    # the same command again: another launcher process
    assert launch(dict(job, TORCHELASTIC_RESTART_COUNT="1"))[0] != first[0]
    # without a local launcher the scheduler's ids alone decide (one daemon per node: no common parent to mix in)
    bare = {"SLURM_JOB_ID": "4711", "MASTER_PORT": "29500"}
    assert launch(bare)[0] == launch(bare)[0]
