"""Link-swap proof of the drop-in boundary against the reference's OWN types.

oracle/_ref/libdropin_amt_f32.so / _f64.so (oracle/Makefile, target `dropin`) are: the reference's module_configure.f90
(its 1 796-field grid_config_rec_type; the routine reads periodic_x, specified, nested, module_configure.f90:434,436,447)
compiled from where it lies + the caller shim oracle/ref_shim.f90 UNMODIFIED (it `use`s module_small_step_em and calls
advance_mu_t with the reference's 48-argument shape, module_small_step_em.f90:7-18) -- with the reference's
module_small_step_em swapped for this repository's (fortran/module_small_step_em.f90: one ISO_C_BINDING call into the HIP
library).  Exactly what a maintainer does to adopt the path: replace one module, relink.  Every committed golden case
(outputs of the reference Fortran itself) must come out bit-equal."""
import ctypes
from pathlib import Path

import numpy as np
import pytest
import torch  # noqa: F401  -- before the HIP library initialises: both must share one HIP runtime (lib.py)

import cases
from conftest import bits_equal

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
pytestmark = pytest.mark.gpu


def _dropin(itemsize):
    p = ROOT / "oracle" / "_ref" / f"libdropin_amt_f{itemsize * 8}.so"
    if not p.exists():
        pytest.skip(f"{p.name} not built (oracle/Makefile `dropin` needs /root/reference: the build container makes it, it travels to the GPU box)")
    L = ctypes.CDLL(str(p))
    return L.ref_advance_mu_t


def test_every_golden_case_through_the_reference_caller_and_config_type(pkg, oracle):
    small = np.load(GOLD / "golden_small.npz")
    keys = sorted({k.rsplit("/", 1)[0] for k in small.files})
    assert len(keys) >= 20
    pkg.load_library()                      # the product library first: the drop-in resolves amt_advance_mu_t_* from it
    done = 0
    for key in keys:
        shape, flag, dtname = key.split("/")
        dt = np.dtype(dtname)
        fn = _dropin(dt.itemsize)
        p = cases.make_case(pkg, shape, flag, dt)
        a = p.args()
        _, args = oracle._marshal(a[:18], a[18:22], a[22:30], a[30], a[31:])     # the checker's ctypes marshalling of the 48 arguments
        fn.restype = None
        fn(*args)
        for n in pkg.synth.OUTPUTS:
            assert bits_equal(p.arrays[n], small[f"{key}/{n}"]), f"{key}/{n}: drop-in module differs from the reference Fortran"
        done += 1
    assert done == len(keys)


def test_the_swapped_library_really_calls_the_hip_path():
    """No Fortran compute path hides in the drop-in: its advance_mu_t is an undefined-symbol import of the HIP library."""
    import subprocess
    p = ROOT / "oracle" / "_ref" / "libdropin_amt_f64.so"
    if not p.exists():
        pytest.skip("not built")
    out = subprocess.run(["nm", "-D", str(p)], capture_output=True, text=True).stdout
    assert " U amt_advance_mu_t_f64" in out and " T ref_advance_mu_t" in out
