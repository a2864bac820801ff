"""Host logic without a GPU: the compute window (module_small_step_em.f90:91-106) as exported by the
C-ABI against a direct Python restatement, and the j-slab split, on random bounds."""
from hypothesis import given, settings, strategies as st


def fortran_window(px, sp, ne, ids, ide, jds, jde, its, ite, jts, jte, kts, kte):
    i_start, i_end = its, min(ite, ide - 1)
    j_start, j_end = jts, min(jte, jde - 1)
    if not px and (sp or ne):
        i_start, i_end = max(its, ids + 1), min(ite, ide - 2)
    if sp or ne:
        j_start, j_end = max(jts, jds + 1), min(jte, jde - 2)
    return i_start, i_end, j_start, j_end, kts, kte - 1


@settings(max_examples=300, deadline=None)
@given(px=st.booleans(), sp=st.booleans(), ne=st.booleans(),
       ids=st.integers(-3, 3), ni=st.integers(1, 50), jds=st.integers(-3, 3), nj=st.integers(1, 50),
       a=st.integers(0, 60), b=st.integers(0, 60), c=st.integers(0, 60), d=st.integers(0, 60), kte=st.integers(1, 70))
def test_compute_window_matches_the_fortran_logic(pkg, px, sp, ne, ids, ni, jds, nj, a, b, c, d, kte):
    ide, jde = ids + ni, jds + nj
    its, ite = sorted((ids + a % (ni + 1), ids + b % (ni + 1)))
    jts, jte = sorted((jds + c % (nj + 1), jds + d % (nj + 1)))
    got = pkg.compute_window(pkg.GridConfig(px, sp, ne), ids, ide, jds, jde, its, ite, jts, jte, 1, kte)
    assert got == fortran_window(px, sp, ne, ids, ide, jds, jde, its, ite, jts, jte, 1, kte)


@settings(max_examples=200, deadline=None)
@given(nj=st.integers(1, 500), world=st.integers(1, 16), sp=st.booleans())
def test_slab_windows_tile_the_unsplit_window(pkg, nj, world, sp):
    """The union of the slabs' compute windows is exactly the unsplit window, without overlap."""
    S = pkg.synth
    if nj < world:
        return
    g = S.domain_bounds(7, 3, nj)
    cfg = pkg.GridConfig(specified=sp)
    whole = pkg.compute_window(cfg, g.ids, g.ide, g.jds, g.jde, g.its, g.ite, g.jts, g.jte, g.kts, g.kte)
    rows = []
    for r in range(world):
        b = S.slab_bounds(g, r, world)
        w = pkg.compute_window(cfg, b.ids, b.ide, b.jds, b.jde, b.its, b.ite, b.jts, b.jte, b.kts, b.kte)
        assert w[:2] == whole[:2]
        rows += list(range(w[2], w[3] + 1))
        assert b.jms <= w[2] - 1 and w[3] + 1 <= b.jme or w[3] < w[2]     # halo rows are in memory
    assert rows == list(range(whole[2], whole[3] + 1))
