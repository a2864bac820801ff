"""What the multi-rank tests share: rank processes on one device, and the checker's UNSPLIT oracle run with inputs that change
every sweep.

Why the inputs change.  What crosses a patch boundary -- rows of v, v_1, t_1, muv, msfvx_inv, columns of u, u_1, t_1, muu, msfuy
(module_small_step_em.f90:143-146, 241-245) -- are pure inputs of advance_mu_t; in WRF advance_uv rewrites u and v before every
call, and the reference re-uploads every array on every call (advance_mu_t_no_async.cu:245-306): always fresh.  With constant
inputs a halo that was delivered ONCE is right for ever, and a stale staging buffer, a pull that races the refresh or an
exchange that silently stops are all invisible from sweep 2 on.  So every multi-rank test gives sweep s (0-based) its own
values of those fields (``synth.refresh_exchanged_inputs``: the generator with seed + s), re-poisons the halos with NaN before
each sweep, and the checker does the same refill on the whole domain before each oracle sweep.  ``AMT_TEST_FAULT`` (csrc/
amt_internal.h) injects exactly those failures; tests/test_gpu_34_halo_freshness.py asserts that each of them turns the check red.
"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SLAB_WORKER = ROOT / "tests" / "workers" / "slab_ipc_rank.py"
GRID_WORKER = ROOT / "tests" / "workers" / "grid_ipc_rank.py"


from conftest import bits_equal  # noqa: E402


def oracle_sweeps(pkg, oracle, patch, seed, sweeps, *, refresh=True, before_each=None, threads=None):
    """``sweeps`` oracle sweeps over the host patch ``patch`` in place; sweep s > 0 first gets the exchanged inputs of seed + s
    (refresh), then ``before_each(patch)`` (loopback tests copy the halo rows / columns by hand there)."""
    S = pkg.synth
    for s in range(sweeps):
        if s and refresh:
            S.refresh_exchanged_inputs(patch, seed, s)
        if before_each is not None:
            before_each(patch)
        if threads and threads > 1:
            oracle.advance_mu_t_omp(*patch.args(), nthreads=threads)
        else:
            oracle.advance_mu_t(*patch.args())
    return patch


def _communicate(procs, what, timeout=420):
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError(f"a {what} hung:\n" + "\n".join(outs))
    return outs


def _rank_env(tag, extra_env, fault):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AMT_TEST_FAULT")}
    env.update(AMT_RENDEZVOUS_NONCE=tag, AMT_SLAB_TRANSPORT="ipc", AMT_IPC_DEVICE_TIMEOUT_S="20", AMT_IPC_TIMEOUT_S="90",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    if fault:
        env["AMT_TEST_FAULT"] = fault
    return env


def run_slab_ranks(tmp_path, world, dims, *, dtype="f64", sweeps=2, overlap=True, specified=False, extra_env=None, seed=11,
                   jitter_us=0, static_inputs=False, fault=None):
    """`world` processes on cuda:0, one j-slab each (tests/workers/slab_ipc_rank.py); returns their outputs (stdout + stderr)."""
    env = _rank_env(f"ipc-{tmp_path.name}", extra_env, fault)
    procs = []
    for r in range(world):
        cmd = [sys.executable, str(SLAB_WORKER), "--rank", str(r), "--world", str(world), "--dir", str(tmp_path), "--dims",
               *map(str, dims), "--dtype", dtype, "--sweeps", str(sweeps), "--seed", str(seed)]
        cmd += [] if overlap else ["--no-overlap"]
        cmd += ["--specified"] if specified else []
        cmd += ["--jitter-us", str(jitter_us)] if jitter_us else []
        cmd += ["--static-inputs"] if static_inputs else []
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = _communicate(procs, "rank")
    assert [p.returncode for p in procs] == [0] * world, "\n".join(outs)
    return outs


def slab_mismatches(pkg, oracle, tmp_path, world, dims, dtype, sweeps, *, specified=False, seed=11, static_inputs=False):
    """[(rank, output name)] whose owned rows differ from the UNSPLIT oracle run (inputs refreshed per sweep unless static)."""
    S = pkg.synth
    np_dtype = np.float64 if dtype == "f64" else np.float32
    gb = S.domain_bounds(*dims, aligned=True)
    want = S.make_patch(gb, pkg.GridConfig(specified=specified), dtype=np_dtype, seed=seed, global_dims=dims)
    oracle_sweeps(pkg, oracle, want, seed, sweeps, refresh=not static_inputs, threads=min(8, os.cpu_count() or 1))
    bad = []
    for r in range(world):
        sb = S.slab_bounds(gb, r, world)
        for n in S.OUTPUTS:
            got = np.load(tmp_path / f"out_{r}_{n}.npy")
            if not bits_equal(got, want.arrays[n][sb.jts - gb.jms: sb.jte + 1 - gb.jms]):
                bad.append((r, n))
    return bad


def run_grid_ranks(tmp_path, pi, pj, dims, *, dtype="f64", sweeps=2, overlap=True, specified=False, align=32, host_wait="1",
                   static_inputs=False, fault=None, extra_env=None):
    env = _rank_env(f"grid-{tmp_path.name}", dict(extra_env or {}, AMT_IPC_HOST_WAIT=host_wait), fault)
    procs = []
    for r in range(pi * pj):
        cmd = [sys.executable, str(GRID_WORKER), "--rank", str(r), "--grid", str(pi), str(pj), "--dir", str(tmp_path), "--dims",
               *map(str, dims), "--dtype", dtype, "--sweeps", str(sweeps), "--align", str(align)]
        cmd += [] if overlap else ["--no-overlap"]
        cmd += ["--specified"] if specified else []
        cmd += ["--static-inputs"] if static_inputs else []
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = _communicate(procs, "rank")
    assert [p.returncode for p in procs] == [0] * (pi * pj), "\n".join(outs)
    return outs


def grid_mismatches(pkg, oracle, tmp_path, pi, pj, dims, dtype, sweeps, specified, align, *, seed=17, static_inputs=False):
    S = pkg.synth
    np_dtype = np.float64 if dtype == "f64" else np.float32
    gb = S.domain_bounds(*dims)
    full = S.make_patch(gb, pkg.GridConfig(specified=specified), dtype=np_dtype, seed=seed, global_dims=dims)
    oracle_sweeps(pkg, oracle, full, seed, sweeps, refresh=not static_inputs)
    bad = []
    for r in range(pi * pj):
        b = S.patch_bounds(gb, r % pi, r // pi, pi, pj, align_elems=align)
        for n in S.OUTPUTS:
            got = np.load(tmp_path / f"out_{r}_{n}.npy")
            want = full.arrays[n][b.jts - gb.jms: b.jte - gb.jms + 1, ..., b.its - gb.ims: b.ite - gb.ims + 1]
            if not bits_equal(got, want):
                bad.append((r, n))
    return bad


def loopback_halos_by_hand(pkg, patch, *, rows=True, columns=False):
    """What a loopback exchange delivers, done by hand on a host patch: row jte+1 of HALO_FROM_ABOVE <- the patch's own row jts,
    row jts-1 of t_1 <- its own row jte; column ite+1 of HALO_FROM_RIGHT <- its own column its, column its-1 of t_1 <- its own
    column ite (owned rows only: a packed column carries the sender's halo-row cells as they were BEFORE this exchange's rows
    landed; the stencil never reads them -- no diagonals)."""
    S, b, w = pkg.synth, patch.bounds, patch.arrays
    jf, jl = b.jts - b.jms, b.jte - b.jms
    cf, cl = b.its - b.ims, b.ite - b.ims
    if rows:
        for n in S.HALO_FROM_ABOVE:
            w[n][jl + 1] = w[n][jf]
        w["t_1"][jf - 1] = w["t_1"][jl]
    if columns:
        for n in S.HALO_FROM_RIGHT:
            w[n][jf:jl + 1, ..., cl + 1] = w[n][jf:jl + 1, ..., cf]
        w["t_1"][jf:jl + 1, ..., cf - 1] = w["t_1"][jf:jl + 1, ..., cl]
