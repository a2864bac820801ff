"""The residency cache of the one-shot host drop-in (header section 1: amt_host_cache_enable / _check /
amt_host_invalidate): an acoustic sub-step loop in which u, v (and the in/out state) change from call to call
while ww_1, u_1, v_1, t_1, ft stay on the device -- bit-identical to the oracle stepping the same loop; a cached
array changed WITHOUT an invalidate is caught by the checksum mode, and is picked up after an invalidate."""
import numpy as np
import pytest

from test_gpu_10_parity import assert_patch_equal

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cache(pkg):
    pkg.host_cache_enable(True, check=True)
    yield pkg
    pkg.host_cache_enable(False, check=False)
    pkg.load_library().amt_host_release()


def _perturb(p, rng, names):
    for n in names:
        a = p.arrays[n]
        a += (rng.standard_normal(a.shape) * 1e-3).astype(a.dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("rows", [None, 7])
def test_sub_step_loop_with_cached_constants_matches_the_oracle(cache, oracle, monkeypatch, dtype, rows):
    pkg = cache
    if rows:
        monkeypatch.setenv("AMT_STREAM_ROWS", str(rows))          # several chunks: the cached rows go up chunk-wise
    b = pkg.synth.domain_bounds(150, 24, 40)
    got = pkg.synth.make_patch(b, pkg.GridConfig(specified=True), dtype=dtype, seed=11)
    want = got.copy()
    rng = np.random.default_rng(5)
    for step in range(5):
        if step:                                                   # advance_uv of the next sub-step: new u, v
            state = rng.bit_generator.state
            _perturb(got, rng, ("u", "v"))
            rng.bit_generator.state = state
            _perturb(want, rng, ("u", "v"))
        pkg.advance_mu_t(*got.args())
        oracle.advance_mu_t(*want.args())
        assert_patch_equal(pkg, got, want, f"cached sub-step {step} ({np.dtype(dtype).name}, rows={rows})")


def test_a_changed_constant_without_invalidate_is_detected_and_with_it_is_used(cache, oracle):
    pkg = cache
    b = pkg.synth.domain_bounds(96, 12, 20)
    got = pkg.synth.make_patch(b, pkg.GridConfig(), dtype=np.float64, seed=3)
    want = got.copy()
    pkg.advance_mu_t(*got.args())
    oracle.advance_mu_t(*want.args())
    assert_patch_equal(pkg, got, want, "first call")
    # a new Runge-Kutta stage rewrites t_1 and ft ...
    for p in (got, want):
        p.arrays["t_1"] += 0.25
        p.arrays["ft"] *= 1.5
    # ... and the caller forgets to say so: the checksum mode refuses
    with pytest.raises(pkg.AmtError) as err:
        pkg.advance_mu_t(*got.args())
    assert err.value.status == 2 and "amt_host_invalidate" in str(err.value)
    # said for one of the two only: the other is still caught
    pkg.host_invalidate(got.arrays["t_1"])
    with pytest.raises(pkg.AmtError) as err:
        pkg.advance_mu_t(*got.args())
    assert "ft" in str(err.value)
    pkg.host_invalidate(None)
    pkg.advance_mu_t(*got.args())
    oracle.advance_mu_t(*want.args())
    assert_patch_equal(pkg, got, want, "after invalidate")


def test_without_the_check_mode_a_stale_constant_is_really_not_uploaded(pkg, oracle):
    """The cache does what it says: with the check off, a changed-but-not-invalidated ft is NOT seen by the
    device (the result is the oracle's on the OLD ft), and is seen after amt_host_invalidate."""
    pkg.host_cache_enable(True, check=False)
    try:
        b = pkg.synth.domain_bounds(64, 10, 12)
        got = pkg.synth.make_patch(b, pkg.GridConfig(), dtype=np.float64, seed=8)
        old = got.copy()
        pkg.advance_mu_t(*got.args())                              # uploads and caches ft
        oracle.advance_mu_t(*old.args())
        got.arrays["ft"] *= 2.0                                    # host changes ft, says nothing
        stale_want = old.copy()                                    # = state after call 1, OLD ft
        fresh_want = old.copy()
        fresh_want.arrays["ft"] *= 2.0
        second = got.copy()
        pkg.advance_mu_t(*got.args())
        oracle.advance_mu_t(*stale_want.args())
        for n in ("t", "ww", "mu"):
            assert np.array_equal(got.arrays[n], stale_want.arrays[n]), n
        pkg.host_invalidate(second.arrays["ft"])                   # other host array, other address: a new key, fresh upload
        pkg.advance_mu_t(*second.args())
        oracle.advance_mu_t(*fresh_want.args())
        for n in ("t", "ww", "mu"):
            assert np.array_equal(second.arrays[n], fresh_want.arrays[n]), n
    finally:
        pkg.host_cache_enable(False)
        pkg.load_library().amt_host_release()


def test_cache_off_is_the_default_and_the_key_follows_the_arrays(pkg, oracle):
    pkg.host_cache_enable(True, check=True)
    keep = []                                                      # live arrays: every patch has its own addresses
    try:
        for seed, dims in ((1, (40, 8, 9)), (2, (40, 8, 9)), (3, (72, 5, 6))):    # other arrays, other extents: never stale data
            b = pkg.synth.domain_bounds(*dims)
            got = pkg.synth.make_patch(b, pkg.GridConfig(nested=True), dtype=np.float32, seed=seed)
            keep.append(got)
            want = got.copy()
            pkg.advance_mu_t(*got.args())
            oracle.advance_mu_t(*want.args())
            assert_patch_equal(pkg, got, want, f"key change seed {seed}")
    finally:
        pkg.host_cache_enable(False)
        pkg.load_library().amt_host_release()
