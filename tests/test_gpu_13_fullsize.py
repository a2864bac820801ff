"""Parity at BASELINE.json's full sizes, anchored on the oracle rather than on cross-kernel agreement
(VERDICT r01 "weak" 6 and "missing" 3):

* configs[2], 4096 x 60 x 4096 fp64 (82 GB resident): more than 10 % of the rows of one sweep are
  recomputed by the oracle -- 64-row j chunks from regenerated inputs, OpenMP j-tiles on all host
  cores -- among them both domain edges and chunks that straddle the workgroups' j-block boundaries.
* configs[4]'s "async H2D/D2H": the streamed one-shot drop-in on a domain that needs many chunks
  (2048 x 80 x 2048 fp32, 13.6 GB of host arrays), pageable and page-locked, against the oracle.
"""
import ctypes
import os
import re
import time

import numpy as np
import pytest

from conftest import bits_equal, slow_note

pytestmark = pytest.mark.gpu


def _host_gb():
    """Host memory this process may really use, GB: MemAvailable, capped by the cgroup's memory.max."""
    avail = 0.0
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) / 1e6
    except OSError:
        pass
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            v = f.read().strip()
        if v != "max":
            used = 0
            try:
                with open("/sys/fs/cgroup/memory.current") as g:
                    used = int(g.read().strip())
            except (OSError, ValueError):
                pass
            avail = min(avail, (int(v) - used) / 1e9)
    except (OSError, ValueError):
        pass
    return avail


def _granted_cores(cap):
    """Threads worth running: the affinity mask capped by the cgroup's CPU quota (beyond it the CFS throttle
    makes every thread slower: profiles/r02_cpu_scaling.md)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def test_tenth_of_the_rows_of_configs2_against_the_oracle(pkg, oracle):
    import torch
    S = pkg.synth
    L = pkg.load_library()
    dims = (4096, 60, 4096)
    b = S.domain_bounds(*dims, aligned=True)
    need = 10 * b.idim * b.kdim * b.jdim * 8 * 1.05
    if torch.cuda.mem_get_info(0)[0] < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of free HBM")
    cfg = pkg.GridConfig(specified=True)
    dev = S.make_patch(b, cfg, dtype=np.float64, seed=4242, device="cuda:0")
    pkg.advance_mu_t(*dev.args())
    torch.cuda.synchronize()
    m = re.search(r"jrows=(\d+)", L.amt_march_last_kernel().decode())
    jrows = int(m.group(1)) if m else 32
    t0 = time.time()
    rows = 64
    # chunk starts: both domain edges, and chunks whose middle is a j-block boundary (first rows of
    # a workgroup's block come out of its prologue, the last ones end its march)
    first_row = 2                                              # specified: j_start = jds + 1
    nblk = -(-(dims[2] - 2) // jrows)                           # uniform blocks of jrows rows from first_row
    bnd = [first_row + jrows * k for k in sorted({1, nblk // 6, nblk // 3, nblk // 2, (2 * nblk) // 3, nblk - 2, nblk - 1}) if 1 <= k < nblk]
    starts = [1, dims[2] - rows + 1] + [x - rows // 2 for x in bnd if x + rows // 2 <= dims[2]]
    k = 0
    while len(starts) < 7:                                     # one-round launches have few block boundaries: add block interiors
        starts.append(first_row + jrows * k + jrows // 2)
        k += 1
    threads = _granted_cores(rows)
    checked = set()
    for jlo in starts:
        jhi = jlo + rows - 1
        sb = b.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
        want = S.make_patch(sb, cfg, dtype=np.float64, seed=4242, global_dims=dims, device="cuda:0").to_host()
        oracle.advance_mu_t_omp(*want.args(), nthreads=threads)
        for n in S.OUTPUTS:
            got = dev.arrays[n][jlo - b.jms: jhi + 1 - b.jms].cpu().numpy()
            assert bits_equal(got, want.arrays[n][1:-1]), f"rows {jlo}..{jhi}: {n} differs from the oracle"
        checked.update(range(jlo, jhi + 1))
    assert len(checked) >= 0.10 * dims[2], len(checked)
    slow_note("full-size rows against the oracle", time.time() - t0, 90)



@pytest.mark.parametrize("dtype,dims,rows_per_block", [(np.float32, (8192, 80, 2048), 1023), (np.float64, (4096, 80, 2048), 1023)])
def test_blocks_whose_row_offsets_pass_2_gib(pkg, oracle, dtype, dims, rows_per_block):
    """The march's per-lane byte offsets are unsigned 32-bit: a block may span up to 4 GiB (amt_march_max_rows).
    80-level rows of 8192 fp32 / 4096 fp64 elements are 2.65 MB, so a 1023-row block ends 2.7 GB from its first row:
    the last rows of both blocks of a tile (and the first ones) against the oracle, bit for bit.  The fp32 launch is the
    launcher's own choice there (one round of 256 blocks); the fp64 one is forced (its shape stays at 64 rows)."""
    import torch
    S = pkg.synth
    L = pkg.load_library()
    b = S.domain_bounds(*dims, aligned=True)
    wbytes = np.dtype(dtype).itemsize
    need = 11.5 * b.idim * b.kdim * b.jdim * wbytes
    if torch.cuda.mem_get_info(0)[0] < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of free HBM")
    cfg = pkg.GridConfig(specified=True)
    seed = 9090
    dev = S.make_patch(b, cfg, dtype=dtype, seed=seed, device="cuda:0")
    forced = dtype == np.float64
    if forced:
        L.amt_march_force_shape(0, 0, 0, -1, 1, rows_per_block, 0)
    try:
        pkg.advance_mu_t(*dev.args())
        torch.cuda.synchronize()
        label = L.amt_march_last_kernel().decode()
    finally:
        if forced:
            L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
    m = re.search(r"jrows=(\d+)", label)
    assert m and int(m.group(1)) == rows_per_block, label
    row_bytes = b.idim * b.kdim * wbytes
    assert (rows_per_block - 1) * row_bytes > 2**31, "the block must reach past 2 GiB"
    rows = 24
    first_row = 2                                               # specified: j_start = jds + 1
    ends = [first_row + rows_per_block - 1, dims[2] - 1]        # last rows of the two blocks of a tile
    starts = [first_row, first_row + rows_per_block] + [e - rows + 1 for e in ends] + [first_row + rows_per_block - 700]
    threads = _granted_cores(rows)
    for jlo in starts:
        jhi = jlo + rows - 1
        sb = b.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
        want = S.make_patch(sb, cfg, dtype=dtype, seed=seed, global_dims=dims, device="cuda:0").to_host()
        oracle.advance_mu_t_omp(*want.args(), nthreads=threads)
        for n in S.OUTPUTS:
            got = dev.arrays[n][jlo - b.jms: jhi + 1 - b.jms].cpu().numpy()
            assert bits_equal(got, want.arrays[n][1:-1]), f"rows {jlo}..{jhi}: {n} differs from the oracle ({label})"
    del dev
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype,ni,nk,dma", [(np.float64, 512, 60, 1), (np.float32, 1024, 80, 1), (np.float64, 512, 60, 0)],
                         ids=["f64-60", "f32-80-level-groups", "f64-60-register-flavour"])
def test_a_block_at_the_offset_limit(pkg, oracle, dtype, ni, nk, dma):
    """The longest block the launcher allows: forced to more rows than the 32-bit offsets span, the block is clipped to
    amt_march_max_rows -- (4 GiB - 40 level rows) / row bytes - 3 -- and its last rows sit within a few rows of the 4 GiB
    mark (fp64: 512 x 60 columns-by-levels rows of 253 760 bytes, 16 900-odd rows per block; fp32: the level-group shape
    of 80 levels): the last rows of that block, the first ones of the next and both domain edges against the oracle."""
    import torch
    S = pkg.synth
    L = pkg.load_library()
    wbytes = np.dtype(dtype).itemsize
    b0 = S.domain_bounds(ni, nk, 8, aligned=True)
    row_bytes = b0.idim * b0.kdim * wbytes
    limit = (2**32 - 40 * b0.idim * wbytes) // row_bytes - 3
    dims = (ni, nk, limit + 400)
    b = S.domain_bounds(*dims, aligned=True)
    assert (b.idim, b.kdim) == (b0.idim, b0.kdim)
    need = 11.5 * b.idim * b.kdim * b.jdim * wbytes
    if torch.cuda.mem_get_info(0)[0] < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of free HBM")
    cfg = pkg.GridConfig(specified=True)
    seed = 1717
    dev = S.make_patch(b, cfg, dtype=dtype, seed=seed, device="cuda:0")
    L.amt_march_force_shape(0, 0, 0, -1, dma, 1 << 30, 0)
    try:
        pkg.advance_mu_t(*dev.args())
        torch.cuda.synchronize()
        label = L.amt_march_last_kernel().decode()
    finally:
        L.amt_march_force_shape(0, 0, 0, -1, 1, 0, 0)
    assert ("true, 16, " in label or "true, 12, " in label) == bool(dma), label
    jrows = int(re.search(r"jrows=(\d+)", label).group(1))
    assert jrows == limit, (label, limit)
    assert (jrows + 2) * row_bytes <= 2**32 < (jrows + 6) * row_bytes + 40 * b.idim * wbytes, "the block should end at the mark"
    rows = 24
    first_row = 2                                               # specified: j_start = jds + 1
    seam = first_row + jrows                                     # first row of the second block
    starts = [1, dims[2] - rows + 1, seam - rows, seam, seam - 3000]
    threads = _granted_cores(rows)
    for jlo in starts:
        jhi = jlo + rows - 1
        sb = b.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
        want = S.make_patch(sb, cfg, dtype=dtype, seed=seed, global_dims=dims, device="cuda:0").to_host()
        oracle.advance_mu_t_omp(*want.args(), nthreads=threads)
        for n in S.OUTPUTS:
            got = dev.arrays[n][jlo - b.jms: jhi + 1 - b.jms].cpu().numpy()
            assert bits_equal(got, want.arrays[n][1:-1]), f"rows {jlo}..{jhi}: {n} differs from the oracle ({label})"
    del dev
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "page-locked"])
def test_streamed_one_shot_on_a_many_chunk_domain(pkg, oracle, pinned):
    """Host arrays in, host arrays out (advance_mu_t_no_async.cu:245-306, 366-390 done with three
    streams): 2048 x 80 x 2048 fp32 goes up and comes down in ~45 chunks of 320 MB."""
    import torch
    S = pkg.synth
    L = pkg.load_library()
    # BASELINE.json configs[4] is 8192 x 80 x 8192 fp32 with "async H2D/D2H": the largest of these that fits half of
    # 70 % of the host memory this process may use (host arrays + the oracle's copy + slack; AMT_STREAM_TEST_DIMS overrides)
    host_gb = _host_gb()
    cands = [(8192, 80, 8192), (8192, 80, 4096), (4096, 80, 4096), (2048, 80, 2048)]
    if os.environ.get("AMT_STREAM_TEST_DIMS"):
        cands = [tuple(int(x) for x in os.environ["AMT_STREAM_TEST_DIMS"].split("x"))]
    dims = None
    for c in cands:
        cb = S.domain_bounds(*c)
        cgb = 10 * cb.idim * cb.kdim * cb.jdim * 4 / 1e9
        if 2.5 * cgb + 8 <= 0.7 * host_gb and torch.cuda.mem_get_info(0)[0] > 1.1e9 * cgb:      # host arrays + the oracle's copy + slack
            dims = c
            break
    if dims is None:
        pytest.skip(f"needs at least {(2.5 * cgb + 8) / 0.7:.0f} GB of host memory (have {host_gb:.0f})")
    print(f"streamed one-shot at {dims} ({'page-locked' if pinned else 'pageable'}), host memory available {host_gb:.0f} GB")
    b = S.domain_bounds(*dims)
    gb = 10 * b.idim * b.kdim * b.jdim * 4 / 1e9
    cfg = pkg.GridConfig(nested=True)
    host = S.make_patch(b, cfg, dtype=np.float32, seed=99, global_dims=dims, device="cuda:0").to_host()
    torch.cuda.empty_cache()
    want = host.copy()
    threads = _granted_cores(64)
    oracle.advance_mu_t_omp(*want.args(), nthreads=threads)
    pins = []
    try:
        if pinned:
            for n in S.RANK3:
                a = host.arrays[n]
                try:
                    pkg.lib.check(L.amt_host_pin(a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
                except pkg.AmtError as e:
                    pytest.skip(f"cannot page-lock {a.nbytes / 1e9:.1f} GB on this host: {e}")
                pins.append(a)
        t0 = time.time()
        pkg.advance_mu_t(*host.args())
        dt = time.time() - t0
    finally:
        for a in pins:
            L.amt_host_unpin(a.ctypes.data_as(ctypes.c_void_p))
        L.amt_host_release()
    for n in S.FIELD_NAMES:
        assert bits_equal(host.arrays[n], want.arrays[n]), n
    print(f"  one-shot call: {dt:.2f} s = {np.prod(dims) / dt / 1e9:.2f} Gcells/s, {gb:.0f} GB of host arrays")
    slow_note("one-shot host call", dt, 60 * max(1.0, gb / 13.6))


def test_configs4_at_its_stated_size_through_the_streamed_host_path(pkg, oracle):
    """BASELINE.json configs[4] names "8192 x 80 x 8192 fp32 ... async H2D/D2H": the one-shot host drop-in on the
    whole 219 GB of host arrays (reference precedent: advance_mu_t_no_async.cu:245-306, 366-390 with the pinned
    host buffers of advance_mu_t_driver.cu:97-167).  The host cannot hold a second copy for the oracle, so the
    result is checked the way the resident full-size tests are: 16-row j chunks -- both domain edges and chunks
    that straddle the call's chunk seams -- recomputed by the oracle from regenerated inputs (the generator is
    index-based), bit for bit, every output array.  Needs ~245 GB of host memory the process may really use
    (MemAvailable capped by the cgroup); skipped with that figure otherwise."""
    import torch
    S = pkg.synth
    L = pkg.load_library()
    dims = (8192, 80, 8192)
    if os.environ.get("AMT_STREAM_TEST_DIMS"):
        dims = tuple(int(x) for x in os.environ["AMT_STREAM_TEST_DIMS"].split("x"))
    b = S.domain_bounds(*dims)
    gb = 10.3 * b.idim * b.kdim * b.jdim * 4 / 1e9
    host_gb = _host_gb()
    if 1.1 * gb + 4 > 0.8 * host_gb:
        pytest.skip(f"needs {(1.1 * gb + 4) / 0.8:.0f} GB of host memory (have {host_gb:.0f}); "
                    f"test_streamed_one_shot_on_a_many_chunk_domain covers the largest size that fits")
    if torch.cuda.mem_get_info(0)[0] < 40e9:
        pytest.skip("needs 40 GB of free HBM for the generator's staging array and the call's workspace")
    cfg = pkg.GridConfig(specified=True)
    seed = 77
    t_gen = time.time()
    arrays = {}
    for name in S.FIELD_NAMES:                                  # one array at a time through the device generator
        t = torch.empty(b.shape(name), dtype=torch.float32, device="cuda:0")
        fa, _ = S._fill_args(b, name, dims)
        pkg.lib.check(L.amt_synth_fill_device(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), S.FIELD_ID[name], 4,
                                              ctypes.c_void_p(t.data_ptr()), ctypes.c_uint64(seed), *fa))
        arrays[name] = t.cpu().numpy()
        del t
        torch.cuda.empty_cache()
    host = S.Patch(b, cfg, arrays, global_dims=dims)
    print(f"configs[4] streamed: {gb:.0f} GB of host arrays generated in {time.time() - t_gen:.0f} s, host memory available {host_gb:.0f} GB")
    rows_per_chunk = int((320 << 20) / (b.idim * b.kdim * 4 * 10) + 1)       # the call's own chunking (amt_oneshot.hip)
    t0 = time.time()
    try:
        pkg.advance_mu_t(*host.args())
    finally:
        L.amt_host_release()
    dt = time.time() - t0
    print(f"  one-shot call (pageable arrays, download thread): {dt:.2f} s = {np.prod(dims) / dt / 1e9:.2f} Gcells/s, "
          f"{rows_per_chunk} rows per chunk, {-(-dims[2] // rows_per_chunk)} chunks")
    rows = 16
    first = 2                                                    # specified: j_start = jds + 1
    nchunk = -(-(dims[2] - 2) // rows_per_chunk)
    seams = [first + rows_per_chunk * k for k in sorted({1, nchunk // 5, nchunk // 2, (4 * nchunk) // 5, nchunk - 1})]
    starts = [1, dims[2] - rows + 1] + [x - rows // 2 for x in seams if rows // 2 < x and x + rows // 2 <= dims[2]]
    threads = _granted_cores(rows)
    for jlo in starts:
        jhi = jlo + rows - 1
        sb = b.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
        want = S.make_patch(sb, cfg, dtype=np.float32, seed=seed, global_dims=dims, device="cuda:0").to_host()
        oracle.advance_mu_t_omp(*want.args(), nthreads=threads)
        for n in S.OUTPUTS:
            got = host.arrays[n][jlo - b.jms: jhi + 1 - b.jms]
            assert bits_equal(np.ascontiguousarray(got), want.arrays[n][1:-1]), f"rows {jlo}..{jhi}: {n} differs from the oracle"
    # nothing outside the window was written: row jds and row jde-1 .. jde of an output still hold the generator's values
    edge = S.make_patch(b.replace(jms=0, jme=2, jts=1, jte=1), cfg, dtype=np.float32, seed=seed, global_dims=dims, device="cuda:0").to_host()
    assert bits_equal(np.ascontiguousarray(host.arrays["t"][0:2]), edge.arrays["t"][0:2])
    slow_note("configs[4] streamed host call", dt, 120)
