#!/usr/bin/env python3
"""Time the one-shot host drop-in (host arrays in, host arrays out -- the PCIe-inclusive figure).

  python profiles/oneshot.py [--ni 1024 --nk 60 --nj 1024 --dtype f64] [--rows 0 16 32 64]

For each --rows value (0 = library default) times amt_advance_mu_t_* with the ten 3-D host
arrays page-locked (streamed pipeline) and, once, with pageable arrays (one piece).  The first
pinned result is compared bit for bit with a device-resident sweep of the same inputs.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ni", type=int, default=1024)
    ap.add_argument("--nk", type=int, default=60)
    ap.add_argument("--nj", type=int, default=1024)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--rows", type=int, nargs="*", default=[0])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-thread", action="store_true", help="pageable arrays: no download thread (one piece)")
    a = ap.parse_args()
    if a.no_thread:
        os.environ["AMT_STREAM_THREAD"] = "0"
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    from wrf_model_cuda_sample_amd import lib
    L = pkg.load_library()
    S = pkg.synth
    dtype = np.float64 if a.dtype == "f64" else np.float32
    b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
    src = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=12345)
    cells = a.ni * a.nk * a.nj
    up = sum(src.arrays[n].nbytes for n in S.FIELD_NAMES)
    down = sum(src.arrays[n].nbytes for n in S.OUTPUTS)

    dev = src.to_device("cuda:0")
    pkg.advance_mu_t(*dev.args())
    torch.cuda.synchronize()
    want = dev.to_host()
    del dev

    def run(p):
        best = None
        for _ in range(a.reps):
            q = p  # in/out arrays evolve; timing does not depend on the values
            t0 = time.perf_counter()
            pkg.advance_mu_t(*q.args())
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best

    out = {"domain": f"{a.ni}x{a.nk}x{a.nj}", "dtype": a.dtype, "h2d_GB": round(up / 1e9, 3), "d2h_GB": round(down / 1e9, 3)}
    out["pageable"] = []
    for rows in a.rows:
        if rows > 0:
            os.environ["AMT_STREAM_ROWS"] = str(rows)
        else:
            os.environ.pop("AMT_STREAM_ROWS", None)
        p = src.copy()
        pkg.advance_mu_t(*p.args())
        same = all(np.array_equal(p.arrays[n].view(np.uint8), want.arrays[n].view(np.uint8)) for n in S.OUTPUTS)
        t = run(p)
        out["pageable"].append({"rows": rows, "ms": round(t * 1e3, 2), "equals_resident": bool(same)})

    p = src.copy()
    pinned = []
    try:
        for n in S.RANK3:
            arr = p.arrays[n]
            lib.check(L.amt_host_pin(arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))
            pinned.append(arr)
        first = True
        out["pinned"] = []
        for rows in a.rows:
            if rows > 0:
                os.environ["AMT_STREAM_ROWS"] = str(rows)
            else:
                os.environ.pop("AMT_STREAM_ROWS", None)
            if first:
                for n in S.FIELD_NAMES:
                    np.copyto(p.arrays[n], src.arrays[n])
                pkg.advance_mu_t(*p.args())
                same = all(np.array_equal(p.arrays[n].view(np.uint8), want.arrays[n].view(np.uint8)) for n in S.OUTPUTS)
                out["pinned_equals_resident"] = bool(same)
                first = False
            t = run(p)
            out["pinned"].append({"rows": rows, "ms": round(t * 1e3, 2), "Mcells_per_s": round(cells / t / 1e6, 1),
                                  "link_GBps_up": round(up / t / 1e9, 1)})
        # the residency cache (amt_host_cache_enable): a sub-step loop -- call 1 uploads everything, the later calls
        # leave ww_1, u_1, v_1, t_1, ft on the device; pinned arrays, library-default chunking
        os.environ.pop("AMT_STREAM_ROWS", None)
        pkg.host_cache_enable(True)
        try:
            for n in S.FIELD_NAMES:
                np.copyto(p.arrays[n], src.arrays[n])
            t0 = time.perf_counter()
            pkg.advance_mu_t(*p.args())
            first_ms = (time.perf_counter() - t0) * 1e3
            same = all(np.array_equal(p.arrays[n].view(np.uint8), want.arrays[n].view(np.uint8)) for n in S.OUTPUTS)
            t = run(p)
            out["pinned_cached"] = {"first_call_ms": round(first_ms, 2), "later_calls_ms": round(t * 1e3, 2),
                                    "Mcells_per_s": round(cells / t / 1e6, 1), "first_call_equals_resident": bool(same),
                                    "h2d_GB_later_calls": round(sum(src.arrays[n].nbytes for n in S.FIELD_NAMES
                                                                    if n not in ("ww_1", "u_1", "v_1", "t_1", "ft", "t_ave", "ww")) / 1e9, 3)}
        finally:
            pkg.host_cache_enable(False)
    finally:
        for arr in pinned:
            lib.check(L.amt_host_unpin(arr.ctypes.data_as(ctypes.c_void_p)))
    # the same loop on pageable arrays
    pkg.host_cache_enable(True)
    try:
        p = src.copy()
        pkg.advance_mu_t(*p.args())
        t = run(p)
        out["pageable_cached"] = {"later_calls_ms": round(t * 1e3, 2), "Mcells_per_s": round(cells / t / 1e6, 1)}
    finally:
        pkg.host_cache_enable(False)
        L.amt_host_release()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
