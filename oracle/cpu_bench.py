#!/usr/bin/env python3
"""oracle/cpu_bench.py -- TEST INFRASTRUCTURE ONLY: one CPU-baseline measurement per process.

bench.py's cpu_baseline leg starts this script once per matrix entry (a child process that never
touches a GPU): a crash, an out-of-memory kill or a timeout of one entry costs that entry, not the
bench line, and the C port (gcc, libgomp) and the Fortran CPU path (amdflang, LLVM OpenMP) never
share a process.

  --impl c                 oracle/oracle_bench.c: the C restatement, -O3 -march=native, OpenMP j-tiles,
                           pages first touched by the thread that computes them
  --impl fortran           oracle/fortran/advance_mu_t_cpu.f90: the build's own Fortran-90 CPU path
                           (fused, i-blocked, OpenMP j-tiles), same harness on the LLVM OpenMP runtime
  --impl reference_nodump  the REFERENCE Fortran compiled -O3 with its five debug dumps cut out
                           (oracle/_ref/libref_nodump_*.so; timing only), one thread
  --impl reference         the reference as shipped, dumps included (to /dev/null), one thread

Prints one JSON record: impl, size, threads, Mcells_s, ms_per_sweep (median), Mcells_s_fastest_sweep, sweeps, fill_s.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", choices=("c", "fortran", "reference_nodump", "reference"))
    ap.add_argument("--prebuild", action="store_true",
                    help="only build the -march=native libraries of THIS machine (oracle/_native/<cpu>/) and exit: "
                         "compile time is never charged to a measurement's budget or timeout")
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f64")
    ap.add_argument("--size", type=int, nargs=3, metavar=("NI", "NK", "NJ"))
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=2.0, help="time budget of the timed sweeps")
    ap.add_argument("--gj0", type=int, default=0, help="global row of the slab's first memory row")
    ap.add_argument("--gnj", type=int, default=0, help="rows of the whole domain (0: the slab is the domain)")
    ap.add_argument("--seed", type=int, default=12345)
    a = ap.parse_args()
    # Idle OpenMP threads spin instead of sleeping (both runtimes; set before either is loaded): the harness runs one short
    # parallel region per sweep, and on the GPU boxes of this pool (16 granted CPUs of 256) waking sleeping workers costs up
    # to half the sweep -- 16 threads, 2 x EPYC 9575F, Fortran path 2.5-2.6 Gcells/s by default, 2.9-3.0 active, 1.9-2.0
    # passive; the C port 1.5-2.9 in any mode (profiles/r04_raw/cpu_threads.txt).  A caller's own setting wins.
    import os
    os.environ.setdefault("OMP_WAIT_POLICY", "active")
    import importlib.util
    spec = importlib.util.spec_from_file_location("amt_oracle", HERE / "oracle.py")
    O = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(O)
    if a.prebuild:
        t0 = time.perf_counter()
        built, failed = [], []
        for what, fn in (("fortran_f64", lambda: O.fortran_lib(8, native=True)), ("fortran_f32", lambda: O.fortran_lib(4, native=True)),
                         ("bench_llvm", O.bench_llvm_lib), ("bench_gcc", O.bench_lib)):
            try:
                fn()
                built.append(what)
            except Exception as e:  # noqa: BLE001  (a compiler that is not there costs its entries, nothing else)
                failed.append(f"{what}: {type(e).__name__}: {str(e)[-200:]}")
        print(json.dumps({"prebuild": built, "failed": failed, "seconds": round(time.perf_counter() - t0, 2),
                          "dir": str(O._native_dir())}), flush=True)
        return
    if not a.impl or not a.size:
        ap.error("--impl and --size are required")
    dtype = np.float64 if a.dtype == "f64" else np.float32
    ni, nk, nj = a.size
    cells = ni * nk * nj
    threads = max(1, min(a.threads, nj))
    fill = None
    if a.impl in ("c", "fortran"):
        # a short probe sets the repetition count for the budget
        probe, _ = O.bench(dtype, ni, nk, min(nj, max(threads, 8)), threads, 2, gj0=a.gj0, gnj=a.gnj or nj, seed=a.seed, impl=a.impl)
        est = max(min(probe), 1e-3) * 1e-3 * nj / min(nj, max(threads, 8))
        reps = int(max(3, min(30, a.seconds / est)))
        ms, fill = O.bench(dtype, ni, nk, nj, threads, reps, gj0=a.gj0, gnj=a.gnj or nj, seed=a.seed, impl=a.impl)
        ms = ms[1:] if len(ms) > 3 else ms
        sweeps = reps
    else:
        import __graft_entry__ as g
        pkg = g.load_package()          # host-side generator only (amt_synth_fill_host); no device is touched
        S = pkg.synth
        b = S.domain_bounds(ni, nk, nj)
        p = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=a.seed, global_dims=(ni, nk, nj))
        fn = O.ref_nodump_advance_mu_t if a.impl == "reference_nodump" else O.ref_advance_mu_t
        threads = 1
        ms = []
        t_end = time.perf_counter() + a.seconds
        while len(ms) < 3 or (time.perf_counter() < t_end and len(ms) < 30):
            t0 = time.perf_counter()
            fn(*p.args())
            ms.append((time.perf_counter() - t0) * 1e3)
        sweeps = len(ms)
    med = float(np.median(ms))
    print(json.dumps({"impl": a.impl, "size": f"{ni}x{nk}x{nj}", "dtype": a.dtype, "threads": threads,
                      "Mcells_s": round(cells / med / 1e3, 2), "ms_per_sweep": round(med, 4), "sweeps": sweeps,
                      "Mcells_s_fastest_sweep": round(cells / float(min(ms)) / 1e3, 2),
                      "fill_s": None if fill is None else round(fill, 3)}), flush=True)


if __name__ == "__main__":
    main()
