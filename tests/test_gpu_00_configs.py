"""Config parity: ONE oracle-anchored test per BASELINE.json config, collected before everything else
(tests/conftest.py orders the files), each well under a minute, so that whatever happens later in the
run every config has been held against the oracle on this box.

  configs[0]  64 x 40 x 64        whole domain, all four flag combinations, both precisions, every kernel
  configs[1]  512 x 60 x 512 fp64 whole domain in ONE launch of the production kernel, all flags (and fp32)
  configs[2]  4096 x 60 x 4096 fp64, resident: > 5 % of the rows recomputed by the oracle
  configs[3]  the same domain as eight j-slabs: three ranks' slabs, computed from their own slab-local arrays
              (global ids..jde, local jms..jme -- what a rank of eight holds after the halo exchange), must
              carry the bits of the whole-domain run
  configs[4]  8192 x 80 x 8192 fp32, resident (219 GB): > 5 % of the rows against the fp32 oracle bit for
              bit, and against the fp64 oracle at the stated tolerance

The oracle is the C restatement of /root/reference/module_small_step_em.f90:7-252 (oracle/, pinned to the
reference's outputs in tests/golden/); bar: bit equality (1e-12 relative, the north_star's tolerance, is met
with zero error).
"""
import os
import time

import numpy as np
import pytest

import cases
from conftest import bits_equal, slow_note

pytestmark = pytest.mark.gpu

REL_TOL_F64 = 1e-12          # BASELINE.json north_star: "matching Fortran to 1e-12 rel"
FP32_VS_FP64_TOL = 2e-5      # configs[4]: per output array, max|fp32 - fp64| <= 2e-5 * max|fp64|


def _cores(cap):
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def _max_rel(a, b):
    """advance_mu_t_driver.f90:288-300."""
    a = a.astype(np.float64).ravel()
    b = b.astype(np.float64).ravel()
    m = np.maximum(np.abs(a), np.abs(b))
    both = (a != 0) & (b != 0)
    rel = np.where(both, np.abs(a - b) / np.where(m == 0, 1, m), m)
    return float(rel.max()) if rel.size else 0.0


def _assert_outputs(pkg, got, want, what, names=None):
    for n in names or pkg.synth.FIELD_NAMES:
        g, w = np.asarray(got.arrays[n]), np.asarray(want.arrays[n])
        if not bits_equal(g, w):
            raise AssertionError(f"{what}: {n} differs from the oracle in {int((g != w).sum())} elements, "
                                 f"max rel {_max_rel(g, w):.3e} (tolerance {REL_TOL_F64:g}; the bar is bit equality)")


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("flag", sorted(cases.FLAG_COMBOS))
def test_configs0_64x40x64_every_kernel(pkg, oracle, torch_mod, flag, dtype):
    host = cases.make_case(pkg, "64x40x64", flag, dtype)
    want = host.copy()
    oracle.advance_mu_t(*want.args())
    for variant in (pkg.VARIANT_MARCH, pkg.VARIANT_COLUMN, pkg.VARIANT_AUTO):
        dev = host.to_device("cuda:0")
        pkg.advance_mu_t(*dev.args(), variant=variant)
        torch_mod.cuda.synchronize()
        _assert_outputs(pkg, dev.to_host(), want, f"configs[0] {flag} variant {variant}")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("flag", sorted(cases.FLAG_COMBOS))
def test_configs1_512x60x512_single_kernel(pkg, oracle, torch_mod, flag, dtype):
    S = pkg.synth
    L = pkg.load_library()
    b = S.domain_bounds(512, 60, 512, aligned=True)
    cfg = pkg.GridConfig(**cases.FLAG_COMBOS[flag])
    dev = S.make_patch(b, cfg, dtype=dtype, seed=2024, device="cuda:0")
    want = dev.to_host()
    oracle.advance_mu_t_omp(*want.args(), nthreads=_cores(16))
    pkg.advance_mu_t(*dev.args(), variant=pkg.VARIANT_MARCH)           # one launch of the production kernel
    torch_mod.cuda.synchronize()
    assert "amt_march_kernel" in L.amt_march_last_kernel().decode()
    got = dev.to_host()
    _assert_outputs(pkg, got, want, f"configs[1] {flag}")
    if dtype == np.float64:
        for n in S.OUTPUTS:
            assert _max_rel(got.arrays[n], want.arrays[n]) <= REL_TOL_F64


def _check_rows_against_oracle(pkg, oracle, dev, b, cfg, dims, dtype, seed, starts, rows, also_fp64=()):
    """Rows jlo..jlo+rows-1 of the resident result against the oracle on regenerated inputs (the generator
    is a function of the global index)."""
    S = pkg.synth
    checked = set()
    threads = _cores(rows)
    for jlo in starts:
        jhi = jlo + rows - 1
        sb = b.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
        want = S.make_patch(sb, cfg, dtype=dtype, seed=seed, global_dims=dims, device="cuda:0").to_host()
        oracle.advance_mu_t_omp(*want.args(), nthreads=threads)
        got = {n: dev.arrays[n][jlo - b.jms: jhi + 1 - b.jms].cpu().numpy() for n in S.OUTPUTS}
        for n in S.OUTPUTS:
            assert bits_equal(got[n], want.arrays[n][1:-1]), f"rows {jlo}..{jhi}: {n} differs from the oracle"
        if jlo in also_fp64:
            w64 = S.make_patch(sb, cfg, dtype=np.float64, seed=seed, global_dims=dims)
            oracle.advance_mu_t_omp(*w64.args(), nthreads=threads)
            for n in S.OUTPUTS:
                ref = w64.arrays[n][1:-1]
                err = np.abs(got[n].astype(np.float64) - ref).max()
                assert err <= FP32_VS_FP64_TOL * np.abs(ref).max(), (jlo, n, err)
        checked.update(range(jlo, jhi + 1))
    return checked


def test_configs2_and_3_4096x60x4096_fp64_resident(pkg, oracle, torch_mod):
    """configs[2]: one sweep of the whole resident domain, 256 + of its 4096 rows (6 %) recomputed by the oracle --
    both domain edges and chunks that straddle workgroup j-block seams.  configs[3] on the one GPU there is: the
    slabs of ranks 0, 3 and 7 of eight, each computed from its OWN slab-local arrays (rows jlo-1..jhi+1 of the
    generator: what the rank holds once its neighbours' halo rows have arrived), must leave the bits of the
    whole-domain run in every output row they own."""
    import re
    torch = torch_mod
    S = pkg.synth
    L = pkg.load_library()
    dims = (4096, 60, 4096)
    b = S.domain_bounds(*dims, aligned=True)
    need = 11.5 * b.idim * b.kdim * b.jdim * 8
    if torch.cuda.mem_get_info(0)[0] < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of free HBM")
    t0 = time.time()
    cfg = pkg.GridConfig(specified=True)
    seed = 4242
    dev = S.make_patch(b, cfg, dtype=np.float64, seed=seed, device="cuda:0")
    pkg.advance_mu_t(*dev.args())
    torch.cuda.synchronize()
    label = L.amt_march_last_kernel().decode()
    assert "amt_march_kernel<double" in label, label
    m = re.search(r"jrows=(\d+)", label)
    jrows = int(m.group(1)) if m else 64
    rows = 64
    nblk = -(-(dims[2] - 2) // jrows)                                    # blocks of jrows rows from row 2 (specified)
    seams = [2 + jrows * k for k in sorted({1, nblk // 2, nblk - 1}) if 1 <= k < nblk]   # j-block boundaries
    starts = [1, dims[2] - rows + 1] + [s_ - rows // 2 for s_ in seams]
    if len(starts) < 4:
        starts.append(dims[2] // 3)
    checked = _check_rows_against_oracle(pkg, oracle, dev, b, cfg, dims, np.float64, seed, starts, rows)
    assert len(checked) >= 0.05 * dims[2], len(checked)
    # configs[3]: j-slabs of eight ranks
    for rank in (0, 3, 7):
        sb = S.slab_bounds(b, rank, 8)
        slab = S.make_patch(sb, cfg, dtype=np.float64, seed=seed, global_dims=dims, device="cuda:0")
        pkg.advance_mu_t(*slab.args())
        torch.cuda.synchronize()
        for n in S.OUTPUTS:
            mine = slab.arrays[n][sb.jts - sb.jms: sb.jte - sb.jms + 1]
            whole = dev.arrays[n][sb.jts - b.jms: sb.jte - b.jms + 1]
            assert torch.equal(mine.view(torch.int64), whole.view(torch.int64)), f"rank {rank} of 8: {n} differs from the whole-domain run"
        del slab
    del dev
    torch.cuda.empty_cache()
    print(f"configs[2]/[3]: {len(checked)} rows against the oracle, 3 slabs of 8 against the whole domain, {time.time() - t0:.0f} s")
    slow_note("configs[2]/[3]", time.time() - t0, 120)


def test_configs4_8192x80x8192_fp32_resident(pkg, oracle, torch_mod):
    """configs[4]: the whole 219 GB domain resident, one launch; 448 of 8192 rows (5.5 %) bit for bit against the
    fp32 oracle, two of the chunks also against the fp64 oracle at the stated tolerance."""
    import re
    torch = torch_mod
    S = pkg.synth
    L = pkg.load_library()
    dims = (8192, 80, 8192)
    b = S.domain_bounds(*dims, aligned=True)
    need = 10.3 * b.idim * b.kdim * b.jdim * 4 + 8e9
    if torch.cuda.mem_get_info(0)[0] < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of free HBM")
    t0 = time.time()
    cfg = pkg.GridConfig(specified=True)
    seed = 77
    dev = S.make_patch(b, cfg, dtype=np.float32, seed=seed, device="cuda:0")
    pkg.advance_mu_t(*dev.args())
    torch.cuda.synchronize()
    label = L.amt_march_last_kernel().decode()
    assert "amt_march_kernel<float" in label, label
    m = re.search(r"jrows=(\d+)", label)
    jrows = int(m.group(1)) if m else 64
    rows = 64
    nblk = -(-(dims[2] - 2) // jrows)
    seams = [2 + jrows * k for k in sorted({1, nblk // 4, nblk // 2, (3 * nblk) // 4, nblk - 2, nblk - 1}) if 1 <= k < nblk]
    assert len(seams) >= 3, (label, seams)
    starts = [1, dims[2] - rows + 1] + [s - rows // 2 for s in seams]
    k = 0
    while len(starts) < 7:                                               # few, long blocks: add block interiors
        starts.append(2 + jrows * k + jrows // 2)
        k += 1
    checked = _check_rows_against_oracle(pkg, oracle, dev, b, cfg, dims, np.float32, seed, starts, rows,
                                         also_fp64=(starts[0], starts[3]))
    assert len(checked) >= 0.05 * dims[2], len(checked)
    del dev
    torch.cuda.empty_cache()
    print(f"configs[4]: {len(checked)} rows against the oracle, {time.time() - t0:.0f} s")
    slow_note("configs[4]", time.time() - t0, 150)
