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
