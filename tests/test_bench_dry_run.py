"""8-rank readiness on CPU: bench.py's own self-launcher with EIGHT ranks -- the target N -- over gloo,
uneven rows, the oracle injected as the compute callable by tests/workers/bench_cpu_shim.py: launcher environment, rendezvous,
slab bounds, poisoned halos + exchange, verification on every rank, barriers, max over ranks, one JSON line, clean teardown."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SHIM = ROOT / "tests" / "workers" / "bench_cpu_shim.py"


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}


def test_eight_ranks_self_launched_on_cpu_uneven_rows():
    t0 = time.time()
    r = subprocess.run([sys.executable, str(SHIM), "--gpus", "8", "--cpu-dry-run", "--ni", "40", "--nk", "6", "--nj", "61",
                        "--steps", "2", "--warmup", "1", "--launch-timeout", "150"],
                       capture_output=True, text=True, env=_env(), timeout=300, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["ranks_seen"] == 8
    assert out["verified_vs_oracle"] is True
    assert out["launched_by"] == "bench.py self-launch"
    rows = out["config"]["rows_per_rank"]
    assert sum(rows) == 61 and len(rows) == 8 and set(rows) == {7, 8}
    assert out["config"]["halo_bytes_per_rank_per_sweep"] > 0
    assert time.time() - t0 < 150


def test_the_dry_run_has_no_compute_of_its_own():
    """bench.py itself (no shim): --cpu-dry-run refuses -- the product has no CPU path to fall back to."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--cpu-dry-run", "--ni", "16", "--nk", "4", "--nj", "8",
                        "--steps", "1", "--warmup", "1", "--launch-timeout", "60"],
                       capture_output=True, text=True, env=_env(), timeout=200, cwd=str(ROOT))
    assert r.returncode != 0
    assert "no compute callable injected" in r.stderr + r.stdout


def test_a_dying_rank_takes_the_eight_down():
    """Teardown at N = 8: one rank exits non-zero before the group forms; the launcher ends the other seven within its grace
    period and returns that rank's code, nobody is left behind."""
    env = _env()
    env["AMT_BENCH_TEST_DIE_RANK"] = "5"
    t0 = time.time()
    r = subprocess.run([sys.executable, str(SHIM), "--gpus", "8", "--cpu-dry-run", "--ni", "24", "--nk", "4", "--nj", "32",
                        "--steps", "1", "--warmup", "1", "--launch-timeout", "120", "--comm-timeout", "60"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=str(ROOT))
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 5 exited with code 7" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 100


# ---- the first-contact ladder: every transport is a rung run in FRESH child processes under a timeout; the
# ---- outcomes a real 8-GPU node can produce on first contact are injected here (bench.py: AMT_BENCH_TEST_RUNG_<NAME>)
def _ladder(extra_env, *extra_args, gpus=3, expect_rc=0, timeout=300):
    env = _env()
    env.update(extra_env)
    r = subprocess.run([sys.executable, str(SHIM), "--gpus", str(gpus), "--cpu-dry-run", "--ni", "40", "--nk", "6", "--nj", "31",
                        "--steps", "2", "--warmup", "2", "--launch-timeout", "200", *extra_args],
                       capture_output=True, text=True, env=env, timeout=timeout, cwd=str(ROOT))
    assert r.returncode == expect_rc, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


def test_ladder_both_transports_are_timed_and_verified_on_new_inputs():
    out, _ = _ladder({})
    assert [x["rung"] for x in out["ladder"]] == ["preflight", "rccl", "ipc"] and all(x["ok"] for x in out["ladder"])
    assert out["value_transport"] == "rccl" and out["value"] == out["transports"]["rccl"]["value"]
    for t in ("rccl", "ipc"):
        rec = out["transports"][t]
        assert rec["ok"] and rec["verified_first_sweep"] is True and rec["verified_later_sweep_after_new_inputs"] is True
        assert rec["value"] > 0 and len(rec["per_rank_achieved_GBps"]) == 3 and rec["ranks_seen"] == 3
    assert out["verified_vs_oracle"] is True
    assert out["roofline"]["aggregate_peak_GBps"] == 3 * 8000.0 and len(out["roofline"]["per_rank_frac"]) == 3


def test_ladder_rccl_init_refuses_then_the_ipc_rung_carries_the_line():
    """ncclCommInitRank returning an error on every rank (what RCCL does when it dislikes the node): the rung fails cleanly, the
    IPC rung runs in fresh processes, `value` is its figure and the line says which transport it is and why."""
    out, err = _ladder({"AMT_BENCH_TEST_RUNG_RCCL": "refuse"})
    assert out["value_transport"] == "ipc" and out["value"] == out["transports"]["ipc"]["value"] and out["verified_vs_oracle"] is True
    assert out["transports"]["rccl"]["ok"] is False and "ncclCommInitRank failed" in out["transports"]["rccl"]["errors"][0]
    assert [x["ok"] for x in out["ladder"]] == [True, False, True]
    assert "rung rccl failed" in err


def test_ladder_a_rank_that_hangs_in_rccl_costs_the_rung_not_the_launch():
    """One rank never comes back from its RCCL set-up (a peer that cannot be reached): its supervisor ends the child by pid after
    --rung-timeout, the other ranks' children -- blocked waiting for it -- go the same way, and every rank goes on to the IPC rung."""
    t0 = time.time()
    out, _ = _ladder({"AMT_BENCH_TEST_RUNG_RCCL": "hang:1"}, "--rung-timeout", "25", "--comm-timeout", "30")   # (a healthy rung on a busy 8-core host needs ~5 s)
    assert time.time() - t0 < 200
    rccl = out["transports"]["rccl"]
    assert rccl["ok"] is False and 1 in rccl["timed_out_ranks"]
    assert out["value_transport"] == "ipc" and out["transports"]["ipc"]["ok"] and out["verified_vs_oracle"] is True


def test_ladder_ipc_handle_refused_rccl_line_unaffected():
    out, _ = _ladder({"AMT_BENCH_TEST_RUNG_IPC": "open_fails"})
    assert out["value_transport"] == "rccl" and out["transports"]["rccl"]["ok"]
    assert out["transports"]["ipc"]["ok"] is False and "hipIpcOpenMemHandle" in out["transports"]["ipc"]["errors"][0]


def test_ladder_nothing_works_the_line_says_so_and_the_exit_code_is_not_zero():
    out, _ = _ladder({"AMT_BENCH_TEST_RUNG_RCCL": "refuse", "AMT_BENCH_TEST_RUNG_IPC": "open_fails"}, expect_rc=5)
    assert out["value"] is None and out["value_transport"] is None
    assert [x["ok"] for x in out["ladder"]] == [True, False, False]


def test_single_transport_request_runs_one_rung():
    out, _ = _ladder({}, "--transport", "ipc", gpus=2)
    assert [x["rung"] for x in out["ladder"]] == ["preflight", "ipc"] and out["value_transport"] == "ipc"


def test_no_rung_child_outlives_the_launch():
    """The children of a rung hold the GPUs.  SIGTERM to the launcher while every rank's RCCL child hangs: the launcher ends the
    supervisors (their process groups), and the children -- started with PR_SET_PDEATHSIG -- go with them; nothing is left behind
    that could keep a device busy for the next job."""
    import re
    import signal
    env = _env()
    env["AMT_BENCH_TEST_RUNG_RCCL"] = "hang"
    p = subprocess.Popen([sys.executable, str(SHIM), "--gpus", "2", "--cpu-dry-run", "--ni", "24", "--nk", "4", "--nj", "16", "--steps", "1",
                          "--warmup", "1", "--launch-timeout", "300", "--rung-timeout", "250"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=str(ROOT))
    pids, buf = [], ""
    t_end = time.time() + 120
    while len(pids) < 2 and time.time() < t_end:
        line = p.stderr.readline()
        buf += line
        pids = [int(x) for x in re.findall(r"rung rccl pid (\d+): hanging", buf)]
    assert len(pids) == 2, buf[-3000:]

    def alive(pid):
        try:
            os.kill(pid, 0)
            return True
        except ProcessLookupError:
            return False
    assert all(alive(q) for q in pids)
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM
    t_end = time.time() + 10
    while any(alive(q) for q in pids) and time.time() < t_end:
        time.sleep(0.2)
    assert not any(alive(q) for q in pids), "a rung child outlived its supervisor"


def test_ladder_under_the_torch_launcher_the_driver_command_line():
    """The driver's own N > 1 command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P <script> --gpus N ... -- the supervisors form their group through the launcher's store, the rung children
    through file stores of their own."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(SHIM), "--gpus", "2", "--cpu-dry-run", "--ni", "40", "--nk", "6", "--nj", "21",
                        "--steps", "2", "--warmup", "2"],
                       capture_output=True, text=True, env=_env(), timeout=400, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["launched_by"] == "external launcher" and out["n_gpus"] == 2 and out["verified_vs_oracle"] is True
    assert out["transports"]["rccl"]["ok"] and out["transports"]["ipc"]["ok"]
