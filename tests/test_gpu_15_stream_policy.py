"""The cache policy of the once-read streams t, ft, ww_1 (profiles/r06_rows4098_nt.md): the launcher
takes the non-temporal instantiation (NTL = 1) where rows are whole 128-byte lines and the plain-load one (NTL = 0) where they
are not -- WRF's own unpadded ims:ime, where neighbouring tiles share the edge line of those streams.  Both are the same
arithmetic: bit-equal to the oracle on either layout, whichever is forced (amt_march_set_stream_policy)."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture()
def policy(pkg):
    L = pkg.load_library()
    yield L.amt_march_set_stream_policy
    L.amt_march_set_stream_policy(-1)


def _run(pkg, oracle, b, dtype, cfg, seed=91):
    import torch
    S = pkg.synth
    host = S.make_patch(b, cfg, dtype=dtype, seed=seed)
    want = host.copy()
    oracle.advance_mu_t(*want.args())
    dev = host.to_device("cuda:0")
    pkg.advance_mu_t(*dev.args(), variant=pkg.VARIANT_MARCH)
    torch.cuda.synchronize()
    label = pkg.load_library().amt_march_last_kernel().decode()
    got = dev.to_host()
    for n in S.OUTPUTS:
        assert bits_equal(got.arrays[n], want.arrays[n]), (n, label)
    return label


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_the_launcher_picks_by_row_length_and_both_policies_are_the_same_bits(pkg, oracle, policy, dtype):
    S = pkg.synth
    cfg = pkg.GridConfig(specified=True)
    padded = S.domain_bounds(200, 30, 24, aligned=True)               # rows of 256 elements: whole lines
    wrf = S.domain_bounds(200, 30, 24)                                # rows of 202 elements: 1616 / 808 bytes, not whole lines
    itemsize = np.dtype(dtype).itemsize
    assert (padded.idim * itemsize) % 128 == 0 and (wrf.idim * itemsize) % 128 != 0
    policy(-1)
    assert ", nt>" in _run(pkg, oracle, padded, dtype, cfg)
    assert ", cached>" in _run(pkg, oracle, wrf, dtype, cfg)
    for forced, tag in ((0, ", cached>"), (1, ", nt>")):
        policy(forced)
        for b in (padded, wrf):
            assert tag in _run(pkg, oracle, b, dtype, cfg)


def test_tall_columns_and_a_patch_window_with_the_cached_policy(pkg, oracle, policy):
    """Level-group shapes (80 and 130 levels) and a patch whose window does not start at the row's first element."""
    S = pkg.synth
    policy(0)
    for nk in (80, 130):
        b = S.domain_bounds(150, nk, 10)
        assert ", cached>" in _run(pkg, oracle, b, np.float64, pkg.GridConfig())
    g = S.domain_bounds(300, 20, 30)
    pb = S.patch_bounds(g, 1, 1, 3, 2)
    S_host = S.make_patch(pb, pkg.GridConfig(nested=True), dtype=np.float32, seed=5, global_dims=(300, 20, 30))
    want = S_host.copy()
    oracle.advance_mu_t(*want.args())
    import torch
    dev = S_host.to_device("cuda:0")
    pkg.advance_mu_t(*dev.args())
    torch.cuda.synchronize()
    got = dev.to_host()
    for n in S.OUTPUTS:
        assert bits_equal(got.arrays[n], want.arrays[n]), n
