"""Timing-based checks, collected after every parity file (tests/conftest.py orders test_gpu_9* last): nothing here
compares bits, and no parity file asserts on the clock."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap", [True, False], ids=["overlap", "no-overlap"])
def test_neighbour_skew_hook_delays_the_sweep(pkg, overlap):
    """amt_slab_set_skew_us holds the communication stream: a 3 ms skew on a 20 us slab must show up almost in full,
    with or without the second stream (lower bound only: a slow box makes both runs slower, not the difference smaller)."""
    import torch
    S = pkg.synth
    gdims = (200, 12, 60)
    b = S.slab_bounds(S.domain_bounds(*gdims, aligned=True), 1, 3)
    times = []
    for skew in (0, 3000):
        dev = S.make_patch(b, pkg.GridConfig(), dtype=np.float64, seed=31, global_dims=gdims, device="cuda:0")
        st = pkg.patch.NativeSlabStepper(dev, 0, 1, pkg.patch.NativeSlabStepper.comm_unique_id(), loopback=True, overlap=overlap)
        st.step(1)
        st.sync()                                                  # connection set-up outside the timing
        st.set_skew_us(skew)
        best = float("inf")
        for _ in range(3):                                         # best of three: one hiccup of the box must not decide
            t0 = time.perf_counter()
            st.step(4)
            st.sync()
            best = min(best, (time.perf_counter() - t0) / 4)
        times.append(best)
        st.close()
    torch.cuda.synchronize()
    assert times[1] > times[0] + 2.0e-3, times
