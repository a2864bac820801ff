"""The launcher's rows-per-workgroup rule (host arithmetic of the HIP library, no GPU): amt_march_rows_for
against a brute-force restatement of the rule in DESIGN.md section 4.2 -- minimise rounds(r) * (r + 0.5) over blocks of at most
the rows the 32-bit offsets span, at most 64 rows for the fp64 shapes with level groups."""
import math
import random

import pytest


def _rule(ntile, nj, cus, max_rows, wbytes=8, hl=1):
    cap = 64 if (wbytes == 8 and hl >= 2) else 1 << 30
    best, pick = None, 1
    for r in range(1, min(nj, max_rows, cap) + 1):
        blocks = ntile * math.ceil(nj / r)
        rounds = math.ceil(blocks / cus)
        cost = rounds * (r + 0.5)
        if best is None or cost < best - 1e-9 or (cost < best + 1e-9 and r > pick):
            best, pick = cost, r
    return pick


@pytest.fixture(scope="module")
def L(pkg):
    return pkg.load_library()


def test_rows_of_the_baseline_configs(L):
    # (tiles, rows, CUs, rows the offsets span) -> rows per workgroup
    assert L.amt_march_rows_for(64, 4094, 256, 1070, 8, 1) == 1024      # configs[2] 4096x60x4096 fp64: 256 blocks, one round
    assert L.amt_march_rows_for(64, 510, 256, 1070, 8, 1) == 128        # configs[3] one j-slab of eight: one round
    assert L.amt_march_rows_for(128, 8190, 256, 1615, 4, 2) == 1365     # configs[4] 8192x80x8192 fp32: 6 blocks per tile, 3 whole rounds
    assert L.amt_march_rows_for(8, 510, 256, 9999, 8, 1) == 16          # configs[1] 512x60x512: 256 blocks of 16 rows
    assert L.amt_march_rows_for(1, 64, 256, 9999, 8, 1) == 1            # one tile: a block per row
    assert L.amt_march_rows_for(128, 2046, 256, 1615, 8, 2) == 64       # 4096x80x2048 fp64, level groups: 64 rows
    assert L.amt_march_rows_for(128, 2046, 256, 1615, 4, 2) == 1023     # 8192x80x2048 fp32: one round


def test_rows_rule_against_brute_force(L):
    rng = random.Random(20260404)
    for _ in range(400):
        ntile = rng.choice([1, 2, 3, 8, 16, 31, 32, 64, 65, 100, 128, 257])
        nj = rng.choice([1, 2, 7, 63, 64, 65, 200, 510, 512, 1000, 2046, 4094, 8190, rng.randint(1, 9000)])
        cus = rng.choice([256, 304, 64, 8])
        max_rows = rng.choice([3, 100, 806, 1070, 100000])
        wbytes, hl = rng.choice([(8, 1), (8, 2), (8, 4), (4, 1), (4, 2)])
        got = L.amt_march_rows_for(ntile, nj, cus, max_rows, wbytes, hl)
        assert got == _rule(ntile, nj, cus, max_rows, wbytes, hl), (ntile, nj, cus, max_rows, wbytes, hl)
        assert 1 <= got <= min(nj, max_rows)
        if wbytes == 8 and hl >= 2:
            assert got <= 64


def test_rows_rule_refuses_empty_questions(L):
    for args in [(0, 100, 256, 1000, 8, 1), (64, 0, 256, 1000, 8, 1), (64, 100, 0, 1000, 8, 1), (64, 100, 256, 0, 8, 1)]:
        assert L.amt_march_rows_for(*args) == 0
