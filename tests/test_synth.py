"""The synthetic-input generator: C-ABI host fill == numpy restatement; device fill == host fill."""
import numpy as np
import pytest

import synth_np
from conftest import bits_equal


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("aligned", [False, True])
def test_host_fill_matches_numpy_restatement(pkg, dtype, aligned):
    S = pkg.synth
    ni, nk, nj = 45, 9, 13
    b = S.domain_bounds(ni, nk, nj, aligned=aligned)
    p = S.make_patch(b, dtype=dtype, seed=99)
    for name in S.FIELD_NAMES:
        want = synth_np.synth(name, 99, (b.idim, b.kdim, b.jdim), (b.ims, b.kms - 1, b.jms),
                              (ni + 2, nk + 1, nj + 2), dtype)
        assert bits_equal(p.arrays[name], want), name


def test_values_do_not_depend_on_padding_or_slab(pkg):
    S = pkg.synth
    ni, nk, nj = 40, 6, 12
    g = S.domain_bounds(ni, nk, nj)
    ga = S.domain_bounds(ni, nk, nj, aligned=True)
    p, pa = S.make_patch(g, seed=5), S.make_patch(ga, seed=5)
    off = g.ims - ga.ims
    for name in S.RANK3 + S.RANK2:
        assert bits_equal(p.arrays[name], np.ascontiguousarray(pa.arrays[name][..., off:off + g.idim])), name
    sb = S.slab_bounds(g, 1, 3)
    ps = S.make_patch(sb, seed=5, global_dims=(ni, nk, nj))
    for name in S.RANK3 + S.RANK2:
        assert bits_equal(ps.arrays[name], np.ascontiguousarray(p.arrays[name][sb.jms - g.jms: sb.jme - g.jms + 1])), name


def test_ranges_keep_divisors_away_from_zero(pkg):
    p = pkg.synth.make_patch(pkg.synth.domain_bounds(64, 10, 64), seed=1)
    for name in ("msfuy", "msfty", "msftx", "msfvx_inv"):
        assert 0.85 < p.arrays[name].min() and p.arrays[name].max() < 1.15
    assert (p.arrays["dnw"] < 0).all()
    assert np.allclose(p.arrays["fnm"] + p.arrays["fnp"], 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_fill_matches_host_fill(pkg, dtype):
    import torch
    S = pkg.synth
    b = S.slab_bounds(S.domain_bounds(70, 11, 40, aligned=True), 1, 3)
    host = S.make_patch(b, dtype=dtype, seed=321, global_dims=(70, 11, 40))
    dev = S.make_patch(b, dtype=dtype, seed=321, global_dims=(70, 11, 40), device="cuda:0")
    torch.cuda.synchronize()
    for name in S.FIELD_NAMES:
        assert bits_equal(dev.arrays[name].cpu().numpy(), host.arrays[name]), name
