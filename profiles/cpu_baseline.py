"""CPU baseline matrix of BASELINE.md section 4 on the GPU box's host cores: the oracle's C port of
the Fortran (gcc -O2, no FMA), 1 thread and OpenMP j-tiles on all available cores, at 64x40x64,
512x60x512 and 4096x60x4096 (fp64; the two small ones also fp32); median of >= 5 calls.
Writes profiles/<tag>_cpu_baseline.json (run through gpurun: python profiles/cpu_baseline.py TAG)."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
pkg, oracle = g.load_package(), g.load_oracle()
S = pkg.synth
cores = len(os.sched_getaffinity(0))
rows = []
for (ni, nk, nj), dtypes in (((64, 40, 64), (np.float32, np.float64)), ((512, 60, 512), (np.float32, np.float64)),
                            ((4096, 60, 4096), (np.float64,))):
    for dt in dtypes:
        p = S.make_patch(S.domain_bounds(ni, nk, nj), pkg.GridConfig(), dtype=dt, seed=12345)
        for threads in (1, min(cores, nj)):
            if threads == 1 and ni * nk * nj > 2e8:
                reps = 2            # ~6 s per call
            else:
                reps = 5
            oracle.advance_mu_t_omp(*p.args(), nthreads=threads)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                oracle.advance_mu_t_omp(*p.args(), nthreads=threads)
                ts.append(time.perf_counter() - t0)
            med = float(np.median(ts))
            cells = ni * nk * nj
            itemsize = np.dtype(dt).itemsize
            rows.append({"domain": f"{ni}x{nk}x{nj}", "dtype": np.dtype(dt).name, "threads": threads,
                         "ms": round(med * 1e3, 3), "Mcells_s": round(cells / med / 1e6, 1),
                         "GB_s_algorithmic": round(itemsize * ni * nj * (11 * nk + 14) / med / 1e9, 1), "calls": reps})
            print(rows[-1], flush=True)
out = {"host_cores_available": cores, "compiler": "gcc -O2 -ffp-contract=off -fopenmp (oracle/Makefile)",
       "code": "oracle/advance_mu_t_oracle.c (C port of module_small_step_em.f90:7-252, no debug dumps)", "rows": rows}
Path(__file__).resolve().parent.joinpath(f"{tag}_cpu_baseline.json").write_text(json.dumps(out, indent=1))
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out", f"{tag}_cpu_baseline.json").write_text(json.dumps(out, indent=1))
