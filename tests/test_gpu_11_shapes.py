"""Every wave shape of AMT_VARIANT_MARCH the library carries -- columns per lane (VW), levels per
lane (KPT), level groups per wave (HL), extra DMA'd inputs (XD), LDS-DMA or register flavour --
forced through amt_march_force_shape and compared bit for bit with the oracle, on level counts that
fill the cell waves and on ragged ones, the aligned resident layout and minimal memory with odd row lengths, all four flag combinations of
module_small_step_em.f90:97-106.  (The launcher's own choice is what every other test runs.)"""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu

# (dtype, vw, kpt, hl, xd, dma, max waves) -- AMT_MARCH_SHAPES of csrc/amt_kernel_march.hip
F64, F32 = np.float64, np.float32
SHAPES = [
    (F64, 1, 2, 1, 0, 1, 16), (F64, 1, 2, 1, 3, 1, 16), (F64, 1, 2, 1, 0, 0, 16),
    (F64, 1, 4, 1, 0, 1, 16), (F64, 1, 4, 1, 1, 1, 16), (F64, 1, 4, 1, 0, 0, 16),
    (F64, 1, 4, 2, 0, 1, 16), (F64, 1, 4, 2, 0, 0, 16),
    (F64, 1, 4, 2, 0, 1, 12), (F64, 1, 4, 2, 1, 1, 12), (F64, 1, 4, 2, 0, 0, 12),
    (F64, 1, 6, 2, 0, 1, 12), (F64, 1, 6, 2, 0, 0, 12),
    (F64, 1, 3, 1, 0, 1, 16), (F64, 1, 3, 2, 0, 1, 16),
    (F64, 1, 2, 2, 0, 1, 16), (F64, 1, 2, 2, 3, 1, 16),
    (F64, 1, 4, 4, 0, 1, 16), (F64, 1, 4, 4, 0, 0, 16), (F64, 1, 4, 4, 0, 1, 12), (F64, 1, 4, 4, 0, 0, 12),
    (F64, 1, 6, 4, 0, 1, 12), (F64, 1, 6, 4, 0, 0, 12),
    (F32, 1, 4, 1, 0, 1, 16), (F32, 1, 4, 1, 3, 1, 16), (F32, 1, 4, 1, 0, 0, 16),
    (F32, 1, 8, 1, 0, 1, 12), (F32, 1, 8, 1, 0, 0, 12),
    (F32, 1, 4, 2, 0, 1, 16), (F32, 1, 4, 2, 0, 0, 16),
    (F32, 1, 8, 2, 0, 1, 12), (F32, 1, 8, 2, 0, 0, 12),
    (F32, 1, 8, 4, 0, 1, 12), (F32, 1, 8, 4, 0, 0, 12),
    (F32, 2, 2, 1, 0, 1, 16), (F32, 2, 2, 1, 3, 1, 16),
    (F32, 2, 4, 1, 0, 1, 16), (F32, 2, 4, 1, 1, 1, 16), (F32, 2, 4, 1, 0, 0, 16),
    (F32, 2, 4, 2, 0, 1, 16), (F32, 2, 4, 2, 0, 0, 16),
    (F32, 2, 4, 2, 0, 1, 12), (F32, 2, 4, 2, 1, 1, 12), (F32, 2, 4, 2, 0, 0, 12),
    (F32, 2, 6, 2, 0, 1, 12), (F32, 2, 6, 2, 0, 0, 12),
    (F32, 2, 3, 1, 0, 1, 16), (F32, 2, 3, 2, 0, 1, 16),
    (F32, 2, 4, 4, 0, 1, 16), (F32, 2, 4, 4, 0, 0, 16), (F32, 2, 4, 4, 0, 1, 12), (F32, 2, 4, 4, 0, 0, 12),
    (F32, 2, 6, 4, 0, 1, 12), (F32, 2, 6, 4, 0, 0, 12),
]


def _id(s):
    return f"{np.dtype(s[0]).name}-vw{s[1]}-kpt{s[2]}-hl{s[3]}-xd{s[4]}-{'dma' if s[5] else 'reg'}-w{s[6]}"


@pytest.fixture()
def force(pkg):
    L = pkg.load_library()
    yield lambda *a: L.amt_march_force_shape(*a)
    L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)


def _levels(kpt, hl):
    lw = kpt * hl
    return sorted({lw, 3 * lw, 3 * lw - 1, 2 * lw + 1, max(1, lw - 1)})      # whole waves and ragged ones


@pytest.mark.parametrize("shape", SHAPES, ids=_id)
def test_forced_shape_matches_oracle(pkg, oracle, force, shape):
    import torch
    dtype, vw, kpt, hl, xd, dma, wm = shape
    S = pkg.synth
    L = pkg.load_library()
    tc = (64 // hl) * vw
    ni = 2 * tc + tc // 2 + 3                       # three tiles, the last partly filled
    flags = [dict(), dict(specified=True), dict(specified=True, periodic_x=True), dict(nested=True)]
    for n, nk in enumerate(_levels(kpt, hl)):
        cfg = pkg.GridConfig(**flags[n % 4])
        for aligned in (True, False):
            # unaligned: minimal memory, an ODD row length (ni + 2) whose last column is the window's
            # i+1 neighbour -- the LDS-DMA takes it as it is (no alignment of the DMA source is needed)
            b = S.domain_bounds(ni, nk, 7, aligned=aligned)
            if not aligned and vw == 2 and not dma:
                # register flavour with two columns per lane: whole pairs from the first tile's column 0 (the window's
                # first column when rows are not whole 128-byte lines, else that column rounded down to a line) to
                # the row end
                def col_lo(bb):
                    line = 128 // np.dtype(dtype).itemsize
                    i_start = pkg.compute_window(cfg, bb.ids, bb.ide, bb.jds, bb.jde, bb.its, bb.ite, bb.jts, bb.jte, bb.kts, bb.kte)[0]
                    i0 = i_start - bb.ims                 # the window's first column (the boundary flags move it)
                    return i0 if bb.idim % line else i0 // line * line
                for _ in range(3):
                    if (b.idim - col_lo(b)) % 2 == 0:
                        break
                    b = b.replace(ime=b.ime + 1)
            host = S.make_patch(b, cfg, dtype=dtype, seed=100 + nk, global_dims=(ni, nk, 7))
            want = host.copy()
            oracle.advance_mu_t(*want.args())
            for jrows in (0, 1, 3):
                force(vw, kpt, hl, xd, dma, jrows, wm)
                dev = host.to_device("cuda:0")
                pkg.advance_mu_t(*dev.args(), variant=pkg.VARIANT_MARCH)
                torch.cuda.synchronize()
                name = L.amt_march_last_kernel().decode()
                # (the label ends with the cache policy of the once-read streams: ", nt>" for rows that are whole 128-byte lines, else ", cached>")
                assert f", {vw}, {kpt}, {hl}, {xd}, FULL, {'true' if dma else 'false'}, {wm}, " in name, name
                assert (", nt>" if (b.idim * np.dtype(dtype).itemsize) % 128 == 0 else ", cached>") in name, name
                got = dev.to_host()
                for f in S.OUTPUTS:
                    assert bits_equal(got.arrays[f], want.arrays[f]), f"{_id(shape)} nk={nk} aligned={aligned} jrows={jrows}: {f} ({name})"


def test_forced_shape_that_cannot_run_fails_loudly(pkg, force):
    import torch
    S = pkg.synth
    b = S.domain_bounds(64, 100, 4, aligned=True)
    dev = S.make_patch(b, pkg.GridConfig(), dtype=np.float64, seed=1, device="cuda:0")
    force(1, 4, 1, 0, 1, 0, 0)                          # 4 levels per lane, one group: at most 60 levels
    with pytest.raises(pkg.AmtError):
        pkg.advance_mu_t(*dev.args(), variant=pkg.VARIANT_MARCH)
    torch.cuda.synchronize()
