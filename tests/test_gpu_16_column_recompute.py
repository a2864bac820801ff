"""The column kernel's RECOMPUTE flavour (csrc/amt_kernel_column.hip): beyond 16 KB of LDS per wave -- 32 fp64 / 64 fp32 levels --
dvdxi(k) is not kept in LDS but evaluated a second time in pass 2, from the same operands with the same expression: the same bits
as the LDS flavour and as the Fortran (module_small_step_em.f90:140-172), at any level count; it is what AUTO runs beyond the march
kernel's 240 / 264 levels (the header's "speed cliff")."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def _run(pkg, host, variant):
    import torch
    dev = host.to_device("cuda:0")
    pkg.advance_mu_t(*dev.args(), variant=variant)
    torch.cuda.synchronize()
    return dev.to_host()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nk", [1, 2, 7, 60, 97, 130, 241, 300])
def test_both_flavours_match_the_oracle_at_every_level_count(pkg, oracle, monkeypatch, dtype, nk):
    S = pkg.synth
    b = S.domain_bounds(150, nk, 5)
    flags = [dict(), dict(specified=True), dict(nested=True, periodic_x=True)][nk % 3]
    host = S.make_patch(b, pkg.GridConfig(**flags), dtype=dtype, seed=300 + nk)
    want = host.copy()
    oracle.advance_mu_t(*want.args())
    for force in ("0", "1", None):                                   # LDS column, recompute, the launcher's choice
        if force is None:
            monkeypatch.delenv("AMT_COLUMN_RECOMPUTE", raising=False)
        else:
            monkeypatch.setenv("AMT_COLUMN_RECOMPUTE", force)
        got = _run(pkg, host, pkg.VARIANT_COLUMN)
        for n in S.OUTPUTS:
            assert bits_equal(got.arrays[n], want.arrays[n]), (n, nk, force)
    monkeypatch.delenv("AMT_COLUMN_RECOMPUTE", raising=False)
    got = _run(pkg, host, pkg.VARIANT_AUTO)                          # beyond 240 / 264 levels: the column kernel, recomputing
    for n in S.OUTPUTS:
        assert bits_equal(got.arrays[n], want.arrays[n]), (n, nk, "auto")
    label = pkg.load_library().amt_march_last_kernel().decode()
    assert ("amt_column_kernel" in label) == (nk > (240 if dtype == np.float64 else 264)), (nk, label)
