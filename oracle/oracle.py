"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front end of the CPU checker for the advance_mu_t path:

* ``advance_mu_t(...)``      -- the plain-C restatement (oracle/advance_mu_t_oracle.c,
  restating /root/reference/module_small_step_em.f90:7-252), same argument list as
  the Fortran routine, numpy arrays updated in place.
* ``advance_mu_t_omp(...)``  -- the same, j-tiled over host threads (the scheme
  sketched in advance_mu_t_driver.f90:175-209); timed by bench.py's cpu_baseline leg.
* ``fortran_advance_mu_t(...)`` -- the build's own Fortran-90 CPU path (oracle/fortran/
  advance_mu_t_cpu.f90: fused, i-blocked, OpenMP j-tiles), the CPU baseline bench.py reports.
* ``ref_advance_mu_t(...)``  -- the REFERENCE Fortran itself (oracle/_ref/, built by
  ``make -C oracle ref`` from the sources under /root/reference; exists only where
  that build has run).  Used to pin the restatement and to generate tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
# AMT_ORACLE_LIBRARY: another build of the same checker (oracle/Makefile `san`: AddressSanitizer + UBSan)
LIB_PATH = Path(os.environ.get("AMT_ORACLE_LIBRARY") or HERE / "liboracle_amt.so")
REF_PATHS = {4: HERE / "_ref" / "libref_amt_f32.so", 8: HERE / "_ref" / "libref_amt_f64.so"}
# timing only: the reference with its five debug dumps cut out (oracle/Makefile, target ref)
REF_NODUMP_PATHS = {4: HERE / "_ref" / "libref_nodump_f32.so", 8: HERE / "_ref" / "libref_nodump_f64.so"}
FORTRAN_CPU_PATHS = {4: HERE / "libamt_cpu_fortran_f32.so", 8: HERE / "libamt_cpu_fortran_f64.so"}

# the five dump files the reference writes into the cwd on every call
# (module_small_step_em.f90:175-189)
_REF_DUMPS = ("muave_before_theta.bin", "mu_before_theta.bin", "mudf_before_theta.bin",
              "muts_before_theta.bin", "ww_before_theta.bin")

# positions in the Fortran argument list
ARRAY_NAMES_A = ("ww", "ww_1", "u", "u_1", "v", "v_1", "mu", "mut", "muave", "muts", "muu", "muv",
                 "mudf", "t", "t_1", "t_ave", "ft", "mu_tend")
SCALAR_NAMES = ("rdx", "rdy", "dts", "epssm")
ARRAY_NAMES_B = ("dnw", "fnm", "fnp", "rdnw", "msfuy", "msfvx_inv", "msftx", "msfty")
INT_NAMES = ("ids", "ide", "jds", "jde", "kde", "ims", "ime", "jms", "jme", "kms", "kme",
             "its", "ite", "jts", "jte", "kts", "kte")


def build(ref: bool | None = None, extras: bool = False) -> None:
    """Compile the C restatement (the checker: gcc only), on request the timing builds and the Fortran CPU
    path (amdclang, amdflang), and the reference build when its sources exist."""
    subprocess.run(["make", "-C", str(HERE), "all"], check=True, capture_output=True)
    if extras:
        subprocess.run(["make", "-C", str(HERE), "extras"], check=True, capture_output=True)
    if ref is None:
        ref = Path("/root/reference/module_small_step_em.f90").exists()
    if ref:
        subprocess.run(["make", "-C", str(HERE), "ref"], check=True, capture_output=True)


_lib = None
_ref_libs: dict = {}


def _sig(real, extra_int=0):
    p = ctypes.c_void_p
    return [p] * 18 + [real] * 4 + [p] * 8 + [ctypes.c_int] * (3 + 17 + extra_int)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build(ref=False)
        L = ctypes.CDLL(str(LIB_PATH))
        for suffix, real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            f = getattr(L, f"oracle_advance_mu_t_{suffix}")
            f.argtypes, f.restype = _sig(real), ctypes.c_int
            g = getattr(L, f"oracle_advance_mu_t_omp_{suffix}")
            g.argtypes, g.restype = _sig(real, 1), ctypes.c_int
        _lib = L
    return _lib


def _native_dir() -> Path:
    """oracle/_native/<hash of this CPU's model and flags>/: builds made -march=native on THIS machine."""
    import hashlib
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln for ln in f if ln.startswith(("model name", "flags"))), "")
            model += "".join(ln for ln in f if ln.startswith("flags"))[:4000]
    except OSError:
        pass
    return HERE / "_native" / hashlib.sha1(model.encode()).hexdigest()[:12]


_fortran_libs: dict = {}


def fortran_lib(itemsize: int, native: bool = False) -> ctypes.CDLL:
    """The Fortran CPU path.  native=False: the in-tree build (-O3 -ffp-contract=off, what the tests
    hold against tests/golden/); native=True: the same source built -march=native on this machine
    (what bench.py times; same bits, tests/test_fortran_cpu.py)."""
    key = (itemsize, native)
    if key not in _fortran_libs:
        override = os.environ.get("AMT_FORTRAN_CPU_DIR")           # A/B builds (profiles/cpu_fortran_tune.sh)
        if override:
            path = Path(override) / FORTRAN_CPU_PATHS[itemsize].name
        elif native:
            out = _native_dir()
            path = out / FORTRAN_CPU_PATHS[itemsize].name
            src = HERE / "fortran" / "advance_mu_t_cpu.f90"
            if not path.exists() or path.stat().st_mtime < src.stat().st_mtime:
                subprocess.run(["make", "-C", str(HERE), f"FCPU_OUT={out}", "FNATIVE=-march=native", "fortran_cpu"],
                               check=True, capture_output=True)
        else:
            path = FORTRAN_CPU_PATHS[itemsize]
            if not path.exists():
                subprocess.run(["make", "-C", str(HERE), "fortran_cpu"], check=True, capture_output=True)
        L = ctypes.CDLL(str(path))
        real = ctypes.c_float if itemsize == 4 else ctypes.c_double
        L.amt_cpu_fortran.argtypes, L.amt_cpu_fortran.restype = _sig(real, 1), ctypes.c_int
        _fortran_libs[key] = L
    return _fortran_libs[key]


_bench_lib = None
_bench_llvm_lib = None


def bench_llvm_lib() -> ctypes.CDLL:
    """oracle_bench.c on the LLVM OpenMP runtime (the Fortran CPU path's), built on this machine."""
    global _bench_llvm_lib
    if _bench_llvm_lib is None:
        out = _native_dir() / "liboracle_bench_llvm.so"
        if not out.exists() or out.stat().st_mtime < (HERE / "oracle_bench.c").stat().st_mtime:
            subprocess.run(["make", "-C", str(HERE), f"BENCH_LLVM_OUT={out}", str(out)], check=True, capture_output=True)
        L = ctypes.CDLL(str(out))
        L.oracle_bench_fn.restype = ctypes.c_int
        L.oracle_bench_fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_long, ctypes.c_long, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                                           ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
        _bench_llvm_lib = L
    return _bench_llvm_lib


def bench_lib() -> ctypes.CDLL:
    """The -O3 -march=native build with the timing driver (oracle_bench.c), compiled on THIS
    machine (the in-tree liboracle_bench.so may come from another CPU)."""
    global _bench_lib
    if _bench_lib is None:
        out = _native_dir() / "liboracle_bench.so"
        if not out.exists() or out.stat().st_mtime < (HERE / "oracle_bench.c").stat().st_mtime:
            subprocess.run(["make", "-C", str(HERE), f"BENCH_OUT={out}", str(out)], check=True, capture_output=True)
        L = ctypes.CDLL(str(out))
        L.oracle_bench.restype = ctypes.c_int
        L.oracle_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_long, ctypes.c_long, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                                        ctypes.POINTER(ctypes.c_double)]
        for suffix, real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            f = getattr(L, f"oracle_advance_mu_t_{suffix}")
            f.argtypes, f.restype = _sig(real), ctypes.c_int
        _bench_lib = L
    return _bench_lib


def bench(dtype, ni, nk, nj, nthreads, reps, *, gj0=0, gnj=None, seed=12345, impl="c"):
    """Timed sweeps on a first-touch-correct NI x NK x NJ domain (a j-slab starting at global row gj0 of
    a domain of gnj rows) of impl "c" (the -O3 C restatement, gcc + libgomp) or "fortran" (the Fortran
    CPU path, amdflang + the LLVM OpenMP runtime, which then also does the first touch).  Returns
    (ms per sweep list, fill seconds).  Do not mix the two impls in one process (two OpenMP runtimes):
    oracle/cpu_bench.py runs each measurement in its own process."""
    ms = (ctypes.c_double * reps)()
    chk, fill = ctypes.c_double(), ctypes.c_double()
    itemsize = np.dtype(dtype).itemsize
    if impl == "fortran":
        fn = ctypes.cast(fortran_lib(itemsize, native=True).amt_cpu_fortran, ctypes.c_void_p)
        rc = bench_llvm_lib().oracle_bench_fn(itemsize, ni, nk, nj, gj0, gnj if gnj is not None else nj, seed,
                                              nthreads, reps, ms, ctypes.byref(chk), ctypes.byref(fill), fn)
    elif impl == "c":
        rc = bench_lib().oracle_bench(itemsize, ni, nk, nj, gj0, gnj if gnj is not None else nj, seed,
                                      nthreads, reps, ms, ctypes.byref(chk), ctypes.byref(fill))
    else:
        raise ValueError(f"unknown impl {impl!r}")
    if rc:
        raise MemoryError(f"oracle_bench: status {rc}")
    if not np.isfinite(chk.value):
        raise ValueError("oracle_bench produced a non-finite checksum")
    return list(ms), fill.value


def have_ref(itemsize: int = 8) -> bool:
    return REF_PATHS[itemsize].exists()


def have_ref_nodump(itemsize: int = 8) -> bool:
    return REF_NODUMP_PATHS[itemsize].exists()


def _ref_lib(itemsize: int, nodump: bool = False) -> ctypes.CDLL:
    key = (itemsize, nodump)
    if key not in _ref_libs:
        real = ctypes.c_float if itemsize == 4 else ctypes.c_double
        L = ctypes.CDLL(str((REF_NODUMP_PATHS if nodump else REF_PATHS)[itemsize]))
        L.ref_advance_mu_t.argtypes, L.ref_advance_mu_t.restype = _sig(real), None
        _ref_libs[key] = L
    return _ref_libs[key]


def _flags(config_flags):
    """config_flags: object with .periodic_x/.specified/.nested, or a mapping, or a 3-tuple
    in the order (periodic_x, specified, nested)."""
    if isinstance(config_flags, (tuple, list)):
        px, sp, ne = config_flags
    elif isinstance(config_flags, dict):
        px, sp, ne = (config_flags.get(k, False) for k in ("periodic_x", "specified", "nested"))
    else:
        px, sp, ne = config_flags.periodic_x, config_flags.specified, config_flags.nested
    return int(bool(px)), int(bool(sp)), int(bool(ne))


def _marshal(arrays_a, scalars, arrays_b, config_flags, ints):
    dt = arrays_a[0].dtype
    if dt not in (np.float32, np.float64):
        raise TypeError(f"unsupported dtype {dt}")
    for a in tuple(arrays_a) + tuple(arrays_b):
        if not isinstance(a, np.ndarray) or a.dtype != dt or not a.flags["F_CONTIGUOUS"] and not a.flags["C_CONTIGUOUS"]:
            raise TypeError("all arrays must be contiguous numpy arrays of one dtype")
    real = ctypes.c_float if dt == np.float32 else ctypes.c_double
    args = [a.ctypes.data_as(ctypes.c_void_p) for a in arrays_a]
    args += [real(float(s)) for s in scalars]
    args += [a.ctypes.data_as(ctypes.c_void_p) for a in arrays_b]
    args += list(_flags(config_flags)) + [int(x) for x in ints]
    return dt, args


def advance_mu_t(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                 t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,
                 msfuy, msfvx_inv, msftx, msfty, config_flags,
                 ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,
                 its, ite, jts, jte, kts, kte, nthreads: int | None = None):
    """CPU restatement; argument list of module_small_step_em.f90:7-18.  Arrays are numpy,
    laid out i-fastest (shape (jdim,kdim,idim) C-order or (idim,kdim,jdim) F-order)."""
    dt, args = _marshal((ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                         t_ave, ft, mu_tend), (rdx, rdy, dts, epssm),
                        (dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty), config_flags,
                        (ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte))
    sfx = "f32" if dt == np.float32 else "f64"
    if nthreads is None:
        rc = getattr(lib(), f"oracle_advance_mu_t_{sfx}")(*args)
    else:
        rc = getattr(lib(), f"oracle_advance_mu_t_omp_{sfx}")(*args, int(nthreads))
    if rc:
        raise ValueError(f"oracle_advance_mu_t: status {rc} (2 = bounds outside what the Fortran defines)")


def advance_mu_t_omp(*args, nthreads: int):
    return advance_mu_t(*args, nthreads=nthreads)


def ref_advance_mu_t(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                     t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,
                     msfuy, msfvx_inv, msftx, msfty, config_flags,
                     ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,
                     its, ite, jts, jte, kts, kte):
    """The reference Fortran routine itself (oracle/_ref).  Runs inside a scratch cwd whose
    five dump-file names point at /dev/null."""
    dt, args = _marshal((ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                         t_ave, ft, mu_tend), (rdx, rdy, dts, epssm),
                        (dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty), config_flags,
                        (ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte))
    L = _ref_lib(dt.itemsize)
    old = os.getcwd()
    with tempfile.TemporaryDirectory(prefix="amt_ref_") as scratch:
        for name in _REF_DUMPS:
            os.symlink("/dev/null", os.path.join(scratch, name))
        os.chdir(scratch)
        try:
            L.ref_advance_mu_t(*args)
        finally:
            os.chdir(old)


def ref_nodump_advance_mu_t(*args48):
    """TIMING ONLY: the reference routine compiled -O3 with its five debug dumps (:175-189) cut out
    (oracle/_ref/libref_nodump_*.so).  Same argument list; never used for parity."""
    (ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1, t_ave, ft, mu_tend,
     rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty, config_flags, *ints) = args48
    dt, args = _marshal((ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1, t_ave, ft, mu_tend),
                        (rdx, rdy, dts, epssm), (dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty), config_flags, ints)
    _ref_lib(dt.itemsize, nodump=True).ref_advance_mu_t(*args)


def fortran_advance_mu_t(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                         t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,
                         msfuy, msfvx_inv, msftx, msfty, config_flags,
                         ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,
                         its, ite, jts, jte, kts, kte, nthreads: int = 1, native: bool = False):
    """The Fortran CPU path (oracle/fortran/advance_mu_t_cpu.f90) on numpy arrays, j-tiled over
    ``nthreads`` OpenMP threads."""
    dt, args = _marshal((ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                         t_ave, ft, mu_tend), (rdx, rdy, dts, epssm),
                        (dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty), config_flags,
                        (ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte))
    rc = fortran_lib(dt.itemsize, native=native).amt_cpu_fortran(*args, int(nthreads))
    if rc:
        raise ValueError(f"amt_cpu_fortran: status {rc} (2 = bounds outside what the Fortran defines)")
