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
