"""The C-ABI library loads and exports every symbol include/amt_advance_mu_t.h declares;
host-side logic (compute window, precondition errors, error reporting) works without a GPU
and nothing silently falls back to a CPU implementation."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

import cases

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "amt_advance_mu_t.h").read_text()


def declared_symbols():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(amt_[a-z0-9_]+)\s*\(", body)))


def test_header_and_binding_table_agree(pkg):
    from wrf_model_cuda_sample_amd import lib
    assert declared_symbols() == sorted(lib.SYMBOLS)


def test_every_declared_symbol_is_exported(pkg):
    L = ctypes.CDLL(str(pkg.library_path()))
    for name in declared_symbols():
        assert hasattr(L, name), f"{name} declared in include/amt_advance_mu_t.h but not exported"


def test_headers_are_plain_c99(tmp_path):
    """include/*.h must be consumable by the reference's C host code (gcc -std=c99)."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "amt_advance_mu_t.h"\n#include "amt_synth.h"\n'
                   'int main(void) { return amt_field_rank(AMT_F_T) == 3 && AMT_OK == 0 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", str(ROOT / "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_version_and_status_strings(pkg):
    L = pkg.load_library()
    assert b"gfx950" in L.amt_version()
    assert L.amt_status_string(0) == b"ok"
    assert L.amt_status_string(2).startswith(b"bounds")


@pytest.mark.parametrize("flags,want", [
    (dict(), (1, 20, 1, 30, 1, 8)),
    (dict(specified=True), (2, 19, 2, 29, 1, 8)),
    (dict(nested=True), (2, 19, 2, 29, 1, 8)),
    (dict(specified=True, periodic_x=True), (1, 20, 2, 29, 1, 8)),
    (dict(periodic_x=True), (1, 20, 1, 30, 1, 8)),
])
def test_compute_window_follows_the_fortran(pkg, flags, want):
    # module_small_step_em.f90:91-106 on a 20x8x30 domain, tile = domain
    got = pkg.compute_window(pkg.GridConfig(**flags), 1, 21, 1, 31, 1, 21, 1, 31, 1, 9)
    assert got == want


def test_compute_window_of_an_interior_tile(pkg):
    got = pkg.compute_window(pkg.GridConfig(specified=True), 1, 21, 1, 31, 5, 9, 1, 12, 1, 9)
    assert got == (5, 9, 2, 12, 1, 8)


@pytest.mark.parametrize("bad", [dict(kts=2), dict(kte=8), dict(kms=2), dict(ims=1), dict(jme=16), dict(ime=16),
                                 dict(jms=1), dict(kme=8)])
def test_preconditions_are_reported_not_fatal(pkg, bad):
    """The reference prints and exit(1)s on a violated precondition
    (advance_mu_t_no_async.cu:82-85); the C-ABI returns AMT_ERR_PRECONDITION."""
    b = pkg.synth.domain_bounds(16, 8, 16).replace(**bad)
    p = pkg.synth.make_patch(b, global_dims=(16, 8, 16))      # arrays sized for the bad bounds
    with pytest.raises(pkg.AmtError) as e:
        pkg.advance_mu_t(*p.args())
    assert e.value.status == 2, str(e.value)


def test_no_cpu_fallback_without_a_device(pkg):
    """On a box without a GPU the compute entry points must fail loudly."""
    L = pkg.load_library()
    if L.amt_device_count() > 0:
        pytest.skip("a device is present")
    p = cases.make_case(pkg, "16x8x16", "none", np.float64)
    before = p.copy()
    with pytest.raises(pkg.AmtError) as e:
        pkg.advance_mu_t(*p.args())
    assert e.value.status in (1, 4)
    for n in pkg.synth.FIELD_NAMES:
        assert np.array_equal(p.arrays[n], before.arrays[n])
    h = ctypes.c_void_p()
    st = L.amt_domain_create(ctypes.byref(h), 8, 0, 0, 0, *p.bounds.as_tuple())
    assert st in (1, 4) and not h.value


def test_product_never_touches_the_oracle():
    """No file of the product package may import, load, link or execute anything of oracle/."""
    pkg_dir = ROOT / "wrf-model-cuda-sample_amd"
    pat = re.compile(r"oracle/|liboracle|load_oracle|import\s+oracle|amt_oracle|oracle\.py|oracle_advance")
    files = [f for ext in ("*.py", "*.hip", "*.h", "*.f90", "*.F90", "*.cpp", "Makefile") for f in pkg_dir.rglob(ext)]
    assert len(files) > 5
    for f in files:
        m = pat.search(f.read_text())
        assert not m, f"{f}: {m.group(0)}"
