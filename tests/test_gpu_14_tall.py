"""Domains with MANY rows per column tile: the launcher then gives a workgroup a long block of rows (one round of
workgroups wherever the launch can be one, profiles/r04_rows.md), the last block of a tile is ragged, and with
AMT_LAUNCH_BESIDE_OTHERS the same launch is planned in at least two rounds.  Whole outputs against the oracle, bit for
bit, on seeded random shapes."""
import json
import os
import re
import time
from pathlib import Path

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu

SEED = int(os.environ.get("AMT_TALL_SEED", "20261003"))
N_CASES = int(os.environ.get("AMT_TALL_CASES", "14"))          # a longer campaign: AMT_TALL_CASES=300


# few tiles x many rows: one round of blocks well beyond 64 rows (the last one of a tile ragged)
LONG = [(2048, 30, 800, np.float64), (4096, 30, 400, np.float64), (1024, 30, 1500, np.float64), (512, 30, 3000, np.float64),
        (1024, 20, 2400, np.float32), (2048, 20, 1777, np.float32)]


def _case(rng, pkg, case):
    S = pkg.synth
    if case < len(LONG):
        ni, nk, nj, dtype = LONG[case]
    else:
        dtype = np.float64 if rng.random() < 0.5 else np.float32
        if rng.random() < 0.6:                      # many columns in all: one round of blocks of more than 64 rows
            ni = int(rng.choice([512, 700, 1024, 1500, 2048, 3000, 4096]))
            nk = int(rng.choice([20, 24, 30, 31]))
            nj = int(rng.choice([510, 777, 1000, 1500, 2047, 3000])) + int(rng.integers(0, 3))
        else:
            ni = int(rng.choice([64, 100, 192, 257, 512, 700, 1024, 1500, 2048]))
            nk = int(rng.choice([4, 8, 13, 20, 30, 40]))
            nj = int(rng.choice([130, 200, 255, 510, 777, 1000, 1500, 2047, 3000]))
        while ni * nk * nj > 50_000_000:
            nj = max(130, nj // 2)
    aligned = bool(rng.integers(0, 2))
    b = S.domain_bounds(ni, nk, nj, aligned=aligned)
    cfg = pkg.GridConfig(periodic_x=bool(rng.integers(0, 2)), specified=bool(rng.integers(0, 2)), nested=bool(rng.integers(0, 2)))
    return b, cfg, dtype, (ni, nk, nj)


def test_tall_domains_match_oracle(pkg, oracle):
    import torch
    torch.cuda.set_device(0)
    rng = np.random.default_rng(SEED)
    S = pkg.synth
    L = pkg.load_library()
    seen_long, seen_beside = 0, 0
    t_start = time.time()
    for case in range(N_CASES):
        b, cfg, dtype, dims = _case(rng, pkg, case)
        beside = case % 2 == 1
        host = S.make_patch(b, cfg, dtype=dtype, seed=500 + case)
        want = host.copy()
        oracle.advance_mu_t_omp(*want.args(), nthreads=8)
        dev = host.to_device("cuda:0")
        pkg.advance_mu_t(*dev.args(), variant=pkg.VARIANT_MARCH | (pkg.LAUNCH_BESIDE_OTHERS if beside else 0))
        torch.cuda.synchronize()
        label = L.amt_march_last_kernel().decode()
        got = dev.to_host()
        for n in S.OUTPUTS:
            assert bits_equal(got.arrays[n], want.arrays[n]), f"case {case} {dims} {np.dtype(dtype).name} {cfg} beside={beside}: {n} ({label})"
        rows = int(re.search(r"jrows=(\d+)", label).group(1))
        seen_long += rows > 64
        print(f"case {case}: {dims} {np.dtype(dtype).name} beside={beside}: {label}")
        seen_beside += beside
        del dev
    assert seen_long >= 3, "the seeded cases should include blocks of more than 64 rows"
    assert seen_beside >= 5
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    if N_CASES > 14 and out.is_dir():
        (out / f"tall_campaign_{N_CASES}.json").write_text(json.dumps({
            "seed": SEED, "cases": N_CASES, "blocks_of_more_than_64_rows": int(seen_long), "launched_beside_others": int(seen_beside),
            "failures": 0, "wall_s": round(time.time() - t_start, 1),
            "what": "tests/test_gpu_14_tall.py: few tiles x many rows (to 4096 columns, 3000 rows, 40 levels), random flags and "
                    "precisions, every other case with AMT_LAUNCH_BESIDE_OTHERS; whole outputs bit-exact against the oracle"}, indent=1) + "\n")


@pytest.mark.parametrize("xchunk", [1, 5, 32])
def test_workgroup_to_block_mapping_knob_keeps_the_bits(pkg, oracle, xchunk):
    """amt_march_set_xchunk only changes WHICH workgroup marches which block: a launch of several rounds with a ragged
    last round (17 tiles x 39 blocks = 663 workgroups) gives the same bits under every mapping."""
    import torch
    torch.cuda.set_device(0)
    S = pkg.synth
    L = pkg.load_library()
    b = S.domain_bounds(1060, 20, 310)
    cfg = pkg.GridConfig(specified=True)
    host = S.make_patch(b, cfg, dtype=np.float64, seed=77)
    want = host.copy()
    oracle.advance_mu_t_omp(*want.args(), nthreads=8)
    L.amt_march_force_shape(0, 0, 0, -1, 1, 8, 0)                   # 8-row blocks: several rounds on any device
    L.amt_march_set_xchunk(xchunk)
    try:
        dev = host.to_device("cuda:0")
        pkg.advance_mu_t(*dev.args(), variant=pkg.VARIANT_MARCH)
        torch.cuda.synchronize()
        label = L.amt_march_last_kernel().decode()
    finally:
        L.amt_march_set_xchunk(0)
        L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
    got = dev.to_host()
    for n in S.OUTPUTS:
        assert bits_equal(got.arrays[n], want.arrays[n]), f"xchunk {xchunk}: {n} ({label})"
