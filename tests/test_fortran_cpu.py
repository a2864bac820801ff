"""The Fortran CPU path (oracle/fortran/advance_mu_t_cpu.f90 -- the build's own fused, i-blocked,
OpenMP-j-tiled restatement; bench.py's cpu_baseline) against tests/golden/ (outputs of the compiled
REFERENCE Fortran) and the C oracle.  Bit-exact, fp32 and fp64; CPU only."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import cases
from conftest import ROOT, bits_equal

GOLD = Path(__file__).resolve().parent / "golden"
DIGESTS = json.loads((GOLD / "golden_digests.json").read_text())


@pytest.mark.parametrize("key", sorted(DIGESTS))
def test_fortran_cpu_path_matches_reference_digests(pkg, oracle, key):
    shape, flag, dtname = key.split("/")
    p = cases.make_case(pkg, shape, flag, np.dtype(dtname))
    oracle.fortran_advance_mu_t(*p.args(), nthreads=1)
    for n in pkg.synth.FIELD_NAMES:
        assert cases.digest(p.arrays[n]) == DIGESTS[key]["outputs"][n], f"{key}: {n} differs from the reference Fortran"


def test_fortran_cpu_path_matches_reference_full_arrays(pkg, oracle):
    small = np.load(GOLD / "golden_small.npz")
    for key in sorted({k.rsplit("/", 1)[0] for k in small.files}):
        shape, flag, dtname = key.split("/")
        p = cases.make_case(pkg, shape, flag, np.dtype(dtname))
        oracle.fortran_advance_mu_t(*p.args(), nthreads=3)
        for n in pkg.synth.OUTPUTS:
            assert bits_equal(p.arrays[n], small[f"{key}/{n}"]), f"{key}/{n}"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_j_tiles_blocks_and_the_native_build_keep_the_bits(pkg, oracle, dtype):
    """i blocks of 1024 columns marching in j (2200 columns = 2 whole blocks + a ragged one; 300 = one ragged
    block), any number of j tiles, and the -march=native build that bench.py times: all the C oracle's bits."""
    for dims in ((300, 17, 23), (2200, 6, 9)):
        b = pkg.synth.domain_bounds(*dims, aligned=True)
        p = pkg.synth.make_patch(b, pkg.GridConfig(nested=True), dtype=dtype, seed=99)
        want = p.copy()
        oracle.advance_mu_t(*want.args())
        for nthreads, native in ((1, False), (4, False), (23, False), (64, False), (3, True)):
            q = p.copy()
            oracle.fortran_advance_mu_t(*q.args(), nthreads=nthreads, native=native)
            for n in pkg.synth.FIELD_NAMES:
                assert bits_equal(q.arrays[n], want.arrays[n]), (dims, nthreads, native, n)


def test_fortran_cpu_path_refuses_undefined_bounds(pkg, oracle):
    p = cases.make_case(pkg, "16x8x16", "none", np.float64)
    with pytest.raises(ValueError):
        oracle.fortran_advance_mu_t(*p.with_bounds(kte=p.bounds.kte - 1).args())


def test_cpu_bench_worker_times_every_implementation():
    """oracle/cpu_bench.py: one measurement per process (the C port on libgomp, the Fortran path on the
    LLVM OpenMP runtime with first touch by the computing threads, the reference without its dumps)."""
    for impl in ("c", "fortran", "reference_nodump"):
        if impl == "reference_nodump" and not (ROOT / "oracle" / "_ref" / "libref_nodump_f64.so").exists():
            continue
        r = subprocess.run([sys.executable, str(ROOT / "oracle" / "cpu_bench.py"), "--impl", impl, "--dtype", "f64",
                            "--size", "48", "6", "10", "--threads", "2", "--seconds", "0.2"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        assert rec["impl"] == impl and rec["Mcells_s"] > 0 and rec["sweeps"] >= 3
        assert rec["threads"] == (1 if impl == "reference_nodump" else 2)
