"""Randomised differential test: random domain sizes, memory paddings (ims/ime, jms/jme, kms/kme),
tile sub-ranges, boundary flags and precisions -- every kernel variant and the one-shot path against
the oracle, bit for bit.  Seeded, so a failure reproduces; AMT_RANDOM_CASES=N runs a longer campaign."""
import os

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("AMT_RANDOM_CASES", "2000"))
SEED = int(os.environ.get("AMT_RANDOM_SEED", "20261002"))


def random_case(pkg, rng):
    S = pkg.synth
    ni = int(rng.choice([1, 2, 3, 31, 63, 64, 65, 127, 129, 257, int(rng.integers(1, 300)), int(rng.integers(1, 700))]))
    nk = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33, 40, 60, 61, int(rng.integers(1, 70)), int(rng.integers(61, 300))]))
    nj = int(rng.choice([1, 2, 3, 5, 9, int(rng.integers(1, 40))]))
    ids, ide, jds, jde, kde = 1, ni + 1, 1, nj + 1, nk + 1
    ims = 1 - int(rng.integers(1, 70))
    ime = ide + int(rng.integers(0, 70))
    jms = 1 - int(rng.integers(1, 4))
    jme = jde + int(rng.integers(0, 4))
    kms = int(rng.choice([1, 1, 0, -2]))
    kme = kde + int(rng.integers(0, 3))
    if rng.random() < 0.4:
        # level counts that fill the cell waves of some wave shape and ragged ones, on rows of a
        # multiple of four elements (every other case has whatever row length falls out)
        nk = int(rng.choice([1, 3, 4, 5, 7, 8, 12, 13, 16, 20, 23, 24, 27, 31, 35, 40, 45, 57, 60, 61, 70, 80, 88, 96, 120, 132, 176, 240]))
        kde, kme = nk + 1, nk + 1 + int(rng.integers(0, 3))
        ime += (-(ime - ims + 1)) % 4
    # tile: the whole domain, or a random sub-tile (as an OpenMP tile / slab would be)
    its, ite, jts, jte = 1, ide, 1, jde
    if rng.random() < 0.5:
        its = int(rng.integers(1, ide + 1)); ite = int(rng.integers(its, ide + 1))
    if rng.random() < 0.5:
        jts = int(rng.integers(1, jde + 1)); jte = int(rng.integers(jts, jde + 1))
    flags = dict(periodic_x=bool(rng.integers(0, 2)), specified=bool(rng.integers(0, 2)), nested=bool(rng.integers(0, 2)))
    b = S.Bounds(ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, 1, kde)
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    return b, pkg.GridConfig(**flags), dtype, (ni, nk, nj)


def test_random_cases_match_oracle(pkg, oracle):
    import torch
    torch.cuda.set_device(0)
    import json
    import time
    from pathlib import Path
    rng = np.random.default_rng(SEED)
    S = pkg.synth
    ran = {"march": 0, "column": 0, "oneshot": 0, "oneshot_cached": 0, "oneshot_deferred": 0}
    t_start = time.time()
    L = pkg.load_library()
    for case in range(N_CASES):
        b, cfg, dtype, dims = random_case(pkg, rng)
        host = S.make_patch(b, cfg, dtype=dtype, seed=1000 + case, global_dims=dims)
        want = host.copy()
        oracle.advance_mu_t(*want.args())
        what = f"case {case}: bounds={b.as_tuple()} flags={cfg} dtype={np.dtype(dtype).name}"
        # one case in five through the register flavour of the march kernel (AMT_MARCH_DMA=0's twin)
        L.amt_march_force_shape(0, 0, 0, -1, 0 if case % 5 == 4 else 1, 0, 0)
        for variant, name in ((pkg.VARIANT_MARCH, "march"), (pkg.VARIANT_COLUMN, "column")):
            dev = host.to_device("cuda:0")
            try:
                pkg.advance_mu_t(*dev.args(), variant=variant)
            except pkg.AmtError as e:
                if variant == pkg.VARIANT_MARCH and e.status == 3:
                    continue                    # no march shape for this level count (AUTO takes the column kernel)
                raise AssertionError(f"{what}: {e}")
            torch.cuda.synchronize()
            got = dev.to_host()
            for n in S.FIELD_NAMES:
                assert bits_equal(got.arrays[n], want.arrays[n]), f"{what}: {name} kernel, {n} differs"
            ran[name] += 1
        if case % 4 == 0:
            one = host.copy()
            pkg.advance_mu_t(*one.args())
            for n in S.FIELD_NAMES:
                assert bits_equal(one.arrays[n], want.arrays[n]), f"{what}: one-shot, {n} differs"
            ran["oneshot"] += 1
        if case % 16 == 8:
            # the one-shot call with the residency cache: two sub-steps, the second one without re-uploading
            # ww_1, u_1, v_1, t_1, ft (checksum mode on: a stale cached array would be refused)
            one, two = host.copy(), want.copy()
            pkg.host_cache_enable(True, check=True)
            try:
                pkg.advance_mu_t(*one.args())
                pkg.advance_mu_t(*one.args())
            finally:
                pkg.host_cache_enable(False)
            oracle.advance_mu_t(*two.args())
            for n in S.FIELD_NAMES:
                assert bits_equal(one.arrays[n], two.arrays[n]), f"{what}: cached one-shot, second sub-step, {n} differs"
            ran["oneshot_cached"] += 1
        if case % 16 == 4:
            # ... and with the outputs deferred as well (r04): two sub-steps, nothing comes down until the fetch; check mode
            # on, so the host arrays hold NaN canaries in between and a silent host write would be refused
            one, two = host.copy(), want.copy()
            pkg.host_cache_enable(True, check=True)
            pkg.host_defer(None, True)
            try:
                pkg.advance_mu_t(*one.args())
                pkg.advance_mu_t(*one.args())
                pkg.host_fetch(None)
            finally:
                pkg.host_defer(None, False)
                pkg.host_cache_enable(False, check=False)
            oracle.advance_mu_t(*two.args())
            for n in S.FIELD_NAMES:
                assert bits_equal(one.arrays[n], two.arrays[n]), f"{what}: deferred one-shot, second sub-step, {n} differs"
            ran["oneshot_deferred"] += 1
    L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
    summary = {"seed": SEED, "cases": N_CASES, "wall_s": round(time.time() - t_start, 1), "ran": ran, "failures": 0,
               "what": "tests/test_gpu_12_random.py: random extents (to 700 columns, 300 levels), paddings, sub-tiles, flags, "
                       "precisions; march (LDS-DMA / register flavour) and column kernels, one-shot, cached and deferred one-shot; "
                       "bit-exact against the oracle"}
    print("random campaign:", json.dumps(summary))
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    if out.is_dir():
        (out / f"random_campaign_{N_CASES}.json").write_text(json.dumps(summary, indent=1) + "\n")
    assert ran["march"] >= N_CASES * 0.8 and ran["column"] >= N_CASES * 0.95


# cases the long campaigns have found (bounds in the order of synth.INT_NAMES, flags, dtype, register flavour?)
REGRESSIONS = [
    # seed 777, case 47199 (r03): fp32 with two columns per lane in the REGISTER flavour on 602-element rows whose
    # window starts at the odd memory column 53: tiles anchored there left the row's last column outside every staged
    # pair and the window's last column read an unwritten t_1(i+1)
    ((1, 549, 1, 4, 34, -52, 549, -1, 5, 0, 34, 1, 549, 1, 4, 1, 34), dict(periodic_x=True, specified=True, nested=True), np.float32, True),
    ((1, 549, 1, 4, 34, -52, 549, -1, 5, 0, 34, 1, 549, 1, 4, 1, 34), dict(periodic_x=True, specified=True, nested=True), np.float32, False),
    ((1, 549, 1, 4, 34, -51, 549, -1, 5, 0, 34, 1, 549, 1, 4, 1, 34), dict(periodic_x=True), np.float32, True),      # odd row length
    ((1, 549, 1, 4, 61, -52, 549, -1, 5, 0, 61, 1, 549, 1, 4, 1, 61), dict(), np.float64, True),
]


@pytest.mark.parametrize("case", range(len(REGRESSIONS)))
def test_cases_found_by_the_campaigns_stay_fixed(pkg, oracle, case):
    import torch
    bounds, flags, dtype, register_flavour = REGRESSIONS[case]
    S = pkg.synth
    L = pkg.load_library()
    b = S.Bounds(*bounds)
    dims = (b.ide - b.ids, b.kde - 1, b.jde - b.jds)
    host = S.make_patch(b, pkg.GridConfig(**flags), dtype=dtype, seed=48199, global_dims=dims)
    want = host.copy()
    oracle.advance_mu_t(*want.args())
    try:
        L.amt_march_force_shape(0, 0, 0, -1, 0 if register_flavour else 1, 0, 0)
        for variant in (pkg.VARIANT_MARCH, pkg.VARIANT_AUTO):
            dev = host.to_device("cuda:0")
            pkg.advance_mu_t(*dev.args(), variant=variant)
            torch.cuda.synchronize()
            got = dev.to_host()
            for n in S.FIELD_NAMES:
                assert bits_equal(got.arrays[n], want.arrays[n]), (case, variant, n, L.amt_march_last_kernel().decode())
    finally:
        L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
