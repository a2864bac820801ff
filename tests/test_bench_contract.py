"""bench.py pieces that do not need a GPU: the algorithmic byte count (SURVEY.md section 8a,
BASELINE.md section 3) and the argument contract."""
import importlib.util
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("amt_bench", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_baseline_md():
    b = _bench()
    assert b.algorithmic_bytes(4096, 60, 4096, 8) == 90_462_748_672          # 90.46 GB / sweep
    assert b.algorithmic_bytes(512, 60, 512, 8) == 1_413_480_448             # 1.413 GB
    assert b.algorithmic_bytes(8192, 80, 8192, 4) == 239_981_297_664         # 240.0 GB
    assert b.algorithmic_bytes(1, 60, 1, 8) == 5392                          # bytes per column, NK = 60
    assert b.HBM_PEAK_GBS == 8000.0


def test_bench_defaults_follow_the_driver_contract():
    b = _bench()
    sys_argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = b.parse()
    finally:
        sys.argv = sys_argv
    assert (a.gpus, a.ni, a.nk, a.nj, a.dtype) == (1, 4096, 60, 4096, "f64")   # BASELINE.json configs[2]
    assert a.steps > 0 and a.warmup >= 0


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "needs a GPU" in (r.stdout + r.stderr)


def test_self_launch_without_gpus_fails_fast_and_loud():
    """`python bench.py --gpus 2` with no launcher starts its own ranks; without a GPU every rank
    refuses (no CPU fallback) and the parent returns their error instead of hanging."""
    import os
    import time
    import torch
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "120"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stdout + r.stderr)
    assert time.time() - t0 < 120


def test_watchdog_returns_raises_and_times_out():
    import time
    b = _bench()
    assert b.watchdog(lambda: 7, 5.0, "quick") == 7
    try:
        b.watchdog(lambda: (_ for _ in ()).throw(ValueError("boom")), 5.0, "raises")
        assert False
    except ValueError as e:
        assert "boom" in str(e)
    t0 = time.time()
    try:
        b.watchdog(lambda: time.sleep(30), 0.3, "ncclCommInitRank stand-in")
        assert False
    except b.StepTimeout as e:
        assert "ncclCommInitRank stand-in" in str(e) and time.time() - t0 < 5


def test_self_launch_takes_hung_ranks_down_on_sigterm_and_on_timeout():
    """A rank that never returns must not outlive the launcher: SIGTERM to the parent, or its own
    --launch-timeout, ends every rank's process group (by pid, never by name)."""
    import os
    import re
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["AMT_BENCH_TEST_HANG"] = "300"

    def alive(pid):
        try:
            os.kill(pid, 0)
            return True
        except ProcessLookupError:
            return False

    # (a) SIGTERM
    p = subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--launch-timeout", "200"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    pids, buf = [], ""
    t_end = time.time() + 60
    while len(pids) < 2 and time.time() < t_end:
        line = p.stderr.readline()
        buf += line
        pids = [int(x) for x in re.findall(r"pid (\d+) hanging", buf)]
    assert len(pids) == 2, buf
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    assert p.returncode == 128 + signal.SIGTERM
    time.sleep(0.5)
    assert not any(alive(q) for q in pids)
    # (b) the launcher's own timeout
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--launch-timeout", "3"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 124 and "timed out after 3 s" in r.stderr and "[rank 1]" in r.stderr
    assert time.time() - t0 < 60
    pids = [int(x) for x in re.findall(r"pid (\d+) hanging", r.stderr)]
    time.sleep(0.5)
    assert len(pids) == 2 and not any(alive(q) for q in pids)


def test_cpu_baseline_failures_cost_an_entry_not_the_line(monkeypatch, tmp_path):
    """Every entry of the CPU-baseline leg is a child process under a timeout; a worker that dies is recorded
    under `errors` and the leg still returns a record (bench.py wraps the whole leg in try/except as well)."""
    b = _bench()
    monkeypatch.setattr(b, "ROOT", tmp_path)                 # no oracle/cpu_bench.py there: every child fails
    out = b.cpu_baseline((64, 8, 32), "f64", 1, 8, 3.0)
    assert out["value"] is None and out["errors"] and out["kind"] == "port"


def test_cpu_baseline_reports_the_fastest_path_and_builds_outside_the_budget():
    """The leg compiles this machine's -march=native libraries in its own child BEFORE the budget starts
    (r03: a cold box's compile ate the first entry's timeout and `value` came back null), and `value` is the
    faster of the Fortran CPU path and the C port on the same slab."""
    b = _bench()
    out = b.cpu_baseline((64, 8, 32), "f64", 1, 8, 6.0)
    assert out["value"] and out["value"] > 0, out.get("errors")
    assert out["value"] == max(x for x in (out["fortran_Mcells_s"], out["port_c_Mcells_s"]) if x)
    assert out["impl"].startswith(("fortran", "port_c")) and "build_seconds_not_in_the_budget" in out
    assert {m["impl"] for m in out["matrix"]} >= {"fortran", "port_c"}
