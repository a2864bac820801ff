#!/usr/bin/env python3
"""bench.py -- advance_mu_t sweeps on N MI355X, one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts N rank processes itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one advance_mu_t sweep (one acoustic sub-step's call) over the whole
4096 x 60 x 4096 fp64 domain (BASELINE.json configs[2]/[3]); inputs are resident in HBM before
the timed region.  With N > 1 the SAME domain is split into N j-slabs (strong scaling), each
rank trades its one-row input halos every sweep (--transport rccl: ncclSend/ncclRecv; ipc: peer copies between
processes, which may share one device) while its interior rows compute.  Default N > 1 stepper: the native C++ runtime behind the C-ABI (amt_slab_*, the path a
Fortran host calls; patch.NativeSlabStepper hands it pointers); --stepper torch runs the same
schedule through torch.distributed P2P ops (patch.SlabStepper) as a cross-check.  Both start from
NaN-poisoned halo rows and are verified against the oracle after the first sweep.

Output keys beyond the driver's contract:
  roofline      algorithmic HBM bytes of one sweep (W*NI*NJ*(11*NK+14))
                divided by the HIP-event time of the kernel launches, against 8 TB/s per GPU
  cpu_baseline  the fastest CPU path -- the build's Fortran-90 restatement or the C port, both j-tiled over the
                host cores -- timed on the WHOLE domain where host memory and the leg's budget allow, else on a bounded
                j-slab sample of it (rank 0, N=1; the record says which)
  placement     the bench state is allocated by amt_domain_create (the product call a Fortran / C host makes once), whose
                default placement sampling keeps the fastest of 4 allocations of the state: `value` is therefore what a
                once-allocating host of the library gets; config.placement_probe_ms lists every allocation's sweep time
                and placement.*_first_allocation says what hipMalloc as it comes would have given
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# the host driver of this pool only supports dmabuf IPC; RCCL's peer-to-peer set-up needs this
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ni", type=int, default=4096)
    ap.add_argument("--nk", type=int, default=60)
    ap.add_argument("--nj", type=int, default=4096)
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 column, 2 march")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--probe-placements", type=int, default=0,
                    help="placement sampling of amt_domain_create, which allocates the bench state: 0 = the library's default "
                         "(AMT_DOMAIN_PLACEMENT_TRIES, 4 allocations of the state, the fastest kept), 1 = the first allocation as "
                         "it comes, K = K allocations; every timing is reported (config.placement_probe_ms)")
    ap.add_argument("--idim-extra", type=int, default=0, help="extra elements of i padding at the end of each row")
    ap.add_argument("--align-elems", type=int, default=32,
                    help="i padding of the resident layout: i = its sits this many elements into a row")
    ap.add_argument("--no-overlap", action="store_true", help="exchange halos before computing (no 2nd stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not re-measure the HBM traffic of a launch with two rocprofv3 --pmc child passes after the "
                         "timed sweeps (N = 1 only; roofline.traffic then is the recorded value of profiles/hbm_traffic.json)")
    ap.add_argument("--no-box-probe", action="store_true",
                    help="skip the in-process streaming ceilings (roofline.box_*) and the clock / power snapshot")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=256,
                    help="j rows of the CPU-baseline slab sample (at least one per granted core; capped by free host memory)")
    ap.add_argument("--cpu-seconds", type=float, default=100.0,
                    help="budget of the whole CPU-baseline leg, fill time included (the whole-domain entry runs only if its estimate fits)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single-GPU projection: do the per-sweep work of ONE rank of an N-slab run (interior + "
                         "edge launches, second stream, halo rows copied device-to-device from local buffers); "
                         "prints a projection line, never the benchmark metric")
    ap.add_argument("--emulate-rank", type=int, default=-1)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo: bring-up mode -- halo rows staged through the host, ranks may share a GPU "
                         "(RCCL refuses that); torch stepper only; never a performance number")
    ap.add_argument("--stepper", choices=("native", "torch"), default="native",
                    help="N > 1: native = amt_slab_* (ncclSend/ncclRecv inside the C++ runtime, what a Fortran "
                         "host calls); torch = torch.distributed P2P ops around the same launches")
    ap.add_argument("--launch-timeout", type=float, default=600.0,
                    help="self-launch (N > 1 without a launcher): give up and end every rank after this many seconds")
    ap.add_argument("--share-gpu", action="store_true",
                    help="N > 1 with fewer GPUs than ranks: map ranks onto the visible devices anyway.  With --transport ipc "
                         "the ranks really run (two processes, one device: a correctness run of the whole N > 1 path, never a "
                         "scaling number); with the RCCL transport this drives the failure path (RCCL refuses two ranks on one "
                         "device: diagnosis, clean non-zero exit of every rank)")
    ap.add_argument("--transport", choices=("rccl", "ipc"), default="rccl",
                    help="N > 1, native stepper: what carries the halo rows -- rccl = ncclSend/ncclRecv (north_star's transport); "
                         "ipc = hipIpcMemHandles + a shared-memory mailbox + copy-engine pulls between the processes of one "
                         "node (no RCCL; no compute unit held while the wire is busy; ranks may share a device)")
    ap.add_argument("--beside-rounds", type=int, default=0,
                    help="N > 1: least rounds of workgroups of a slab's interior launch (amt_march_set_beside; 0 = library default 2)")
    ap.add_argument("--beside-reserve", type=int, default=0,
                    help="N > 1: compute units every round of the interior launch leaves free for the exchange and the edge rows "
                         "(profiles/r05_slab_ab.md: 4 rounds / 16 units make the sweep flat in the neighbours' lateness)")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="N > 1 plumbing without GPUs (tests/test_bench_dry_run.py): the ranks hold CPU tensors over gloo and run "
                         "the same launcher, slabs, halo exchange, verification, reductions and teardown; the compute callable is "
                         "NOT in this file -- the caller injects it (bench.DRY_RUN_COMPUTE); prints a line marked dry_run, never "
                         "a measurement")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="N > 1: most seconds a rank waits in the communicator set-up (ncclCommInitRank) and in the "
                         "first halo exchange before it ends itself with a diagnosis")
    return ap.parse_args()


def gpu_state_smi():
    """Clocks, power and power cap as rocm-smi reports them (a child process; called BEFORE this process
    touches the GPU, so it is the idle state of the box -- the clocks under load come from sysfs, below)."""
    import subprocess
    out = {}
    for flag, key in (("--showclocks", "clocks"), ("--showpower", "power"), ("--showmaxpower", "power_cap"),
                      ("--showperflevel", "perf_level"), ("--showtemp", "temperature")):
        try:
            r = subprocess.run(["rocm-smi", flag, "--json"], capture_output=True, text=True, timeout=20)
            d = json.loads(r.stdout)
            out[key] = d.get("card0", d)
        except Exception as e:  # noqa: BLE001
            out[key] = f"unavailable ({type(e).__name__})"
    return out


def gpu_state_sysfs(index=0):
    """sclk / mclk / fclk levels (the starred one is current) and socket power from sysfs: cheap enough to
    read while kernels are in flight."""
    out = {}
    cards = sorted(Path("/sys/class/drm").glob("card[0-9]*/device/pp_dpm_sclk"))
    if not cards:
        return None
    dev = cards[min(index, len(cards) - 1)].parent
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
        try:
            cur = [ln.strip() for ln in (dev / name).read_text().splitlines() if "*" in ln]
            out[name[7:]] = cur[0] if cur else None
        except OSError:
            pass
    for hw in dev.glob("hwmon/hwmon*/power1_average"):
        try:
            out["power_W"] = int(hw.read_text()) / 1e6
            out["power_cap_W"] = int((hw.parent / "power1_cap").read_text()) / 1e6
        except (OSError, ValueError):
            pass
    return out or None


def box_ceilings(pkg, stream, device, nbytes=4 << 30):
    """The box's own streaming rates, in this process, on `stream`: a tuned 16-byte-per-lane copy
    (nbytes read + nbytes written) and a read-only sweep (amt_calib_stream_rate).  GB/s, best of three
    rounds of four launches each; plus the sysfs clocks sampled while the copies are in flight."""
    import ctypes
    import torch
    L = pkg.load_library()
    src = torch.empty(nbytes // 8, dtype=torch.float64, device=device).fill_(1.0)
    dst = torch.empty_like(src)
    h = ctypes.c_void_p(stream.cuda_stream)
    out = {}
    under_load = None
    for mode, key in ((0, "box_copy_GBps"), (1, "box_read_GBps")):
        def launch():
            pkg.lib.check(L.amt_calib_stream_rate(h, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()),
                                                  ctypes.c_size_t(nbytes), mode))
        launch()
        torch.cuda.synchronize()
        best = float("inf")
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(4 if rnd else 40):          # the first round is long enough to read the clocks under load
                launch()
            e1.record(stream)
            if rnd == 0 and mode == 0:
                time.sleep(0.02)
                under_load = gpu_state_sysfs(device.index or 0)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (4 if rnd else 40))
        out[key] = round((2 if mode == 0 else 1) * nbytes / best / 1e6, 1)
    del src, dst
    torch.cuda.empty_cache()
    return out, under_load


def measure_traffic(a, seconds=150.0):
    """HBM bytes of one launch ON THIS BOX, in this run: two child runs of this same script (two sweeps each) under
    `rocprofv3 --kernel-trace --kernel-include-regex amt_ --pmc FETCH_SIZE` / `... WRITE_SIZE` -- separate passes, the
    program directly after `--`, as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- and the gfx950 correction of
    profiles/README.md (FETCH_SIZE counts half of a streamed read).  Returns (read_bytes, write_bytes, note) or raises;
    the caller has released its own arrays first.  Each pass is its own process group under a timeout."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("this process is itself running under a profiler: no nested rocprofv3 passes")
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="amt_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--output-format", "csv", "--kernel-trace", "--kernel-include-regex", "amt_march|amt_column",
               "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, str(Path(__file__).resolve()),
               "--ni", str(a.ni), "--nk", str(a.nk), "--nj", str(a.nj), "--dtype", a.dtype, "--variant", str(a.variant),
               "--seed", str(a.seed), "--align-elems", str(a.align_elems), "--idim-extra", str(a.idim_extra),
               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-verify", "--no-box-probe",
               "--probe-placements", "1", "--no-traffic"]
        p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp",
                             env=dict(os.environ, TMPDIR="/tmp"), start_new_session=True)
        try:
            p.wait(timeout=seconds)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            p.wait()
            shutil.rmtree(out, ignore_errors=True)
            raise RuntimeError(f"the {counter} pass did not finish in {seconds:.0f} s")
        vals = []
        for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") == counter and "amt_" in r.get("Kernel_Name", ""):
                        vals.append(float(r["Counter_Value"]))
        shutil.rmtree(out, ignore_errors=True)
        if p.returncode != 0 or not vals:
            raise RuntimeError(f"the {counter} pass gave no data (exit {p.returncode})")
        got[counter] = (sum(vals) / len(vals), len(vals))
    rd = 2.0 * got["FETCH_SIZE"][0] * 1024.0
    wr = got["WRITE_SIZE"][0] * 1024.0
    note = (f"measured in this run on this box: child runs of this command under rocprofv3 --kernel-trace --pmc FETCH_SIZE and "
            f"--pmc WRITE_SIZE (separate passes, {got['FETCH_SIZE'][1]} + {got['WRITE_SIZE'][1]} launches; KiB per launch "
            f"{got['FETCH_SIZE'][0]:.0f} / {got['WRITE_SIZE'][0]:.0f}); HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 "
            f"calibration of profiles/README.md")
    return rd, wr, note


def ramp_clocks(pkg, stream, tensor, max_seconds=3.0):
    """Make sure the GPU is out of its idle power state before the warm-up sweeps, and record that it is: the
    verification of the first sweep is a second or two of CPU work during which the GPU idles, and W = 5 warm-up sweeps
    last only 75 ms.  A read-only stream over one of the state's own arrays (nothing is written) runs in batches of
    ~50 ms until two consecutive batches agree to 1 % (at least four, at most max_seconds).  On the boxes seen so far the
    first batch already streams at the last one's rate (no ramp), and the 20 timed sweeps that follow are flat to 0.3 %
    (`per_sweep_ms`); one earlier line whose mean was 16.44 ms against a median of 15.37 had no per-sweep record to say
    which sweeps were slow -- now every line has.  (sysfs's sclk is NOT a witness: it reads `S: 100Mhz` in the middle of
    a 6.9 TB/s stream on this pool.)  Returns what it saw, for the line."""
    import ctypes
    import torch
    L = pkg.load_library()
    nb = tensor.numel() * tensor.element_size() // 16 * 16
    if nb < (1 << 20):
        return None
    sink = torch.zeros(8, dtype=torch.float64, device=tensor.device)
    h = ctypes.c_void_p(stream.cuda_stream)
    per_batch = max(4, int(0.05 / max(nb / 6.0e12, 1e-6)))            # ~50 ms at 6 TB/s
    rates, t_begin = [], time.perf_counter()
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(per_batch):
            pkg.lib.check(L.amt_calib_stream_rate(h, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(tensor.data_ptr()),
                                                  ctypes.c_size_t(nb), 1))
        e1.record(stream)
        e1.synchronize()
        rates.append(per_batch * nb / e0.elapsed_time(e1) / 1e6)
        steady = len(rates) >= 4 and abs(rates[-1] - rates[-2]) <= 0.01 * rates[-1] and abs(rates[-2] - rates[-3]) <= 0.01 * rates[-2]
        if steady or time.perf_counter() - t_begin > max_seconds:
            break
    # one more batch, and the clocks as sysfs shows them in the middle of it (steady state under load)
    for _ in range(per_batch):
        pkg.lib.check(L.amt_calib_stream_rate(h, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(tensor.data_ptr()),
                                              ctypes.c_size_t(nb), 1))
    time.sleep(0.015)
    clocks = gpu_state_sysfs(tensor.device.index or 0)
    torch.cuda.synchronize(tensor.device)
    return {"batches": len(rates), "seconds": round(time.perf_counter() - t_begin, 3), "first_batch_GBps": round(rates[0], 1),
            "last_batch_GBps": round(rates[-1], 1), "slowest_batch_GBps": round(min(rates), 1), "clocks_at_the_end": clocks}


def algorithmic_bytes(ni, nk, nj, itemsize):
    """Compulsory HBM traffic of one sweep (SURVEY.md section 8a / BASELINE.md section 3)."""
    return itemsize * ni * nj * (11 * nk + 14)


def verify_first_sweep(pkg, oracle, dev, gb, dims, dtype, seed, rank_rows):
    """After exactly one sweep from fresh inputs: recompute a few 3-row j-slabs with the oracle
    from regenerated inputs and compare bit for bit (size-independent parity check)."""
    S = pkg.synth
    b = dev.bounds
    jlo_own, jhi_own = rank_rows
    cand = sorted({jlo_own, max(jlo_own, jhi_own - 2), (jlo_own + jhi_own) // 2})
    for jlo in cand:
        jhi = min(jlo + 2, jhi_own)
        sb = gb.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
        want = S.make_patch(sb, dev.config, dtype=dtype, seed=seed, global_dims=dims)
        oracle.advance_mu_t_omp(*want.args(), nthreads=min(3, jhi - jlo + 1))
        for n in S.OUTPUTS:
            got = dev.arrays[n][jlo - b.jms: jhi + 1 - b.jms].cpu().numpy()
            w = want.arrays[n][1: 1 + (jhi - jlo + 1)]
            if not np.array_equal(got.view(np.uint8), w.view(np.uint8)):
                return False, f"rows {jlo}..{jhi} of {n} differ from the oracle"
    return True, ""


def fp32_error_vs_fp64(pkg, oracle, dev, gb, dims, seed, rank_rows):
    """BASELINE.json configs[4]: the fp32 run judged against the fp64 Fortran (oracle in fp64 on a
    3-row slab of regenerated fp64 inputs).  Returns max over the outputs of max|fp32-fp64| / max|fp64|."""
    S = pkg.synth
    b = dev.bounds
    jlo = (rank_rows[0] + rank_rows[1]) // 2
    jhi = min(jlo + 2, rank_rows[1])
    sb = gb.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
    want = S.make_patch(sb, dev.config, dtype=np.float64, seed=seed, global_dims=dims)
    oracle.advance_mu_t_omp(*want.args(), nthreads=min(3, jhi - jlo + 1))
    worst = 0.0
    for n in S.OUTPUTS:
        got = dev.arrays[n][jlo - b.jms: jhi + 1 - b.jms].cpu().numpy().astype(np.float64)
        w = want.arrays[n][1: 1 + (jhi - jlo + 1)]
        worst = max(worst, float(np.abs(got - w).max() / np.abs(w).max()))
    return worst


def host_cores():
    """Threads worth running: the CPUs this process may use, capped by the cgroup's CPU quota (the GPU
    boxes of this pool show 256 logical CPUs and grant 16 CPUs of time: beyond 16 threads the CFS
    throttle makes a sweep slower, 2.7 Gcells/s at 16 threads against 0.25 at 256, profiles/r02_cpu_scaling.md)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} logical CPUs visible"
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                note += f", cgroup cpu.max grants {q}"
                n = q
    except (OSError, ValueError):
        pass
    return n, note


def mem_available_bytes():
    try:
        for ln in Path("/proc/meminfo").read_text().splitlines():
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) * 1024
    except (OSError, ValueError, IndexError):
        pass
    return None


def cpu_baseline(dims, dtype_name, seed, rows, seconds):
    """SURVEY.md section 8(d) / BASELINE.md section 4, timed on this box's host cores in the same run:

      fortran           the build's own Fortran-90 CPU path (oracle/fortran/advance_mu_t_cpu.f90: same
                        48-argument signature, fused single pass over j, i-blocked, no dumps, OpenMP over
                        j-tiles as sketched in advance_mu_t_driver.f90:175-209; amdflang -O3 -march=native
                        -ffp-contract=off; bit-equal to tests/golden/) -- `value` is this one on all granted cores
      port_c            the C restatement the parity tests use as the checker (gcc -O3 -march=native)
      reference_fortran_compute_only
                        the REFERENCE routine itself compiled -O3 with its five debug dumps
                        (module_small_step_em.f90:175-189) cut out; one thread, as the reference runs it
      reference_fortran_incl_dumps   the reference as shipped (dumps to /dev/null): an I/O number, for context

    at 64x40x64, 512x60x512 and a j-slab of the bench domain sized by --cpu-rows and the host's free memory,
    on one thread and on all the cores the process is granted (count stated).  Every entry is its own child
    process (oracle/cpu_bench.py; no GPU is touched there) under a timeout: one that fails costs its entry,
    never the bench line.  Bounded: `seconds` is the budget of the whole leg, fill time included."""
    import subprocess
    ni, nk, nj = dims
    cores, quota_note = host_cores()
    itemsize = 8 if dtype_name == "f64" else 4
    avail = mem_available_bytes()
    row_bytes = 10.5 * (ni + 2) * (nk + 1) * itemsize                  # ten 3-D arrays + the 2-D ones
    slab_rows = max(cores, min(nj, rows))
    if avail:
        slab_rows = max(1, min(slab_rows, int(0.3 * avail / row_bytes)))
    matrix, errors = [], []
    worker = str(ROOT / "oracle" / "cpu_bench.py")
    # The -march=native builds of THIS machine (oracle/_native/<cpu>/: the Fortran CPU path in both precisions, the two
    # harnesses) are compiled first, by their own child under its own timeout: on a cold box that is 20-40 s of amdflang /
    # amdclang / gcc, which belongs neither to the leg's budget nor to any entry's timeout (r03: the first entry's 32 s
    # timeout ended inside the compile on the driver's box and the line went out with value null).
    t_build = time.perf_counter()
    prebuild = None
    try:
        r = subprocess.run([sys.executable, worker, "--prebuild"], capture_output=True, text=True, timeout=600)
        prebuild = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
        if r.returncode != 0:
            errors.append(f"prebuild: exit {r.returncode}: {r.stderr.strip()[-300:]}")
        elif prebuild and prebuild.get("failed"):
            errors.extend(f"prebuild: {x}" for x in prebuild["failed"])
    except Exception as e:  # noqa: BLE001
        errors.append(f"prebuild: {type(e).__name__}: {str(e)[-300:]}")
    build_seconds = round(time.perf_counter() - t_build, 1)
    t_leg = time.perf_counter()


    def run(impl, shape, threads, what="", gj0=0, gnj=0, share=1.0, sweep_seconds=None, timeout=None):
        left = seconds - (time.perf_counter() - t_leg)
        if left <= 0.5:
            errors.append(f"{impl} {shape} x{threads}: skipped, the leg's {seconds:.0f} s budget is spent")
            return None
        budget = sweep_seconds if sweep_seconds else max(0.3, min(left, seconds * share / 12.0))
        cmd = [sys.executable, worker, "--impl", impl, "--dtype", dtype_name, "--size", *map(str, shape),
               "--threads", str(threads), "--seconds", f"{budget:.2f}", "--seed", str(seed), "--gj0", str(gj0), "--gnj", str(gnj)]
        try:
            # (no ORACLE_BENCH_FILL_THREADS for the one-thread entries: on a two-socket host the pages a parallel
            # fill touches land on both sockets and the single compute thread then reads half its data remotely:
            # 151 against 229 Mcells/s for the same code on 2 x EPYC 9575F)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout or max(45.0, 4 * left))
            if r.returncode != 0:
                raise RuntimeError(f"exit {r.returncode}: {r.stderr.strip()[-300:]}")
            rec = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001  (an OOM kill, a timeout, a compiler that is not there ...)
            errors.append(f"{impl} {'x'.join(map(str, shape))} x{threads}: {type(e).__name__}: {str(e)[-300:]}")
            return None
        name = {"c": "port_c", "fortran": "fortran", "reference_nodump": "reference_fortran_compute_only",
                "reference": "reference_fortran_incl_dumps"}[impl]
        rec = {"impl": name, "size": rec["size"] + (f" ({what})" if what else ""), "threads": rec["threads"],
               "Mcells_s": rec["Mcells_s"], "ms_per_sweep": rec["ms_per_sweep"], "sweeps": rec["sweeps"], "fill_s": rec["fill_s"],
               "Mcells_s_fastest_sweep": rec.get("Mcells_s_fastest_sweep")}
        matrix.append(rec)
        return rec

    slab_shape = (ni, nk, slab_rows)
    gj0 = max(0, (nj - slab_rows) // 2)
    slab_what = "j-slab of the bench domain, first touch by the computing thread"
    slab_f = run("fortran", slab_shape, cores, slab_what, gj0, nj, share=3.0)      # the two candidates for `value` first
    slab_c = run("c", slab_shape, cores, slab_what, gj0, nj, share=2.0)
    one = run("fortran", (512, 60, 512), 1)
    run("fortran", (512, 60, 512), cores)
    run("fortran", (64, 40, 64), 1)
    run("fortran", (64, 40, 64), cores)
    have_ref = (ROOT / "oracle" / "_ref" / f"libref_nodump_{dtype_name}.so").exists()
    if have_ref:
        run("reference_nodump", (512, 60, 512), 1)
        run("reference_nodump", (64, 40, 64), 1)
    run("c", (512, 60, 512), 1)
    run("c", (512, 60, 512), cores)
    if (ROOT / "oracle" / "_ref" / f"libref_amt_{dtype_name}.so").exists():
        run("reference", (64, 40, 64), 1)
    # The whole bench domain where the host allows ("the largest size host RAM allows"):
    # MemAvailable at least twice the domain, and the estimate -- the slab entry's fill and sweep times scaled by the rows --
    # inside what is left of the leg's budget.  Pages are first touched by the threads that compute on them, like the slab.
    # The slab sample stays the fallback; the record says which one `value` is.
    full, full_skipped = None, None
    domain_bytes = row_bytes * (nj + 2)
    best_slab = max((c for c in (slab_f, slab_c) if c), key=lambda c: c["Mcells_s"], default=None)
    if slab_rows >= nj:
        full_skipped = "the slab sample IS the whole domain"
    elif not avail or avail < 2 * domain_bytes:
        full_skipped = f"MemAvailable {(avail or 0) / 2**30:.0f} GiB is less than twice the domain's {domain_bytes / 2**30:.0f} GiB"
    elif not best_slab:
        full_skipped = "no slab measurement to size it by"
    else:
        scale = nj / slab_rows
        est = (best_slab["fill_s"] or 0.0) * scale + 5 * best_slab["ms_per_sweep"] * 1e-3 * scale + 3.0
        left = seconds - (time.perf_counter() - t_leg)
        if est > left - 2.0:
            full_skipped = f"estimated {est:.0f} s (fill + 5 sweeps) does not fit the {left:.0f} s left of --cpu-seconds {seconds:.0f}"
        else:
            full_impl = "fortran" if best_slab is slab_f else "c"
            full = run(full_impl, (ni, nk, nj), cores, "the whole bench domain, first touch by the computing threads", 0, nj,
                       sweep_seconds=max(1.0, 4 * best_slab["ms_per_sweep"] * 1e-3 * scale), timeout=max(60.0, 3 * est))
            if not full:
                full_skipped = "the full-size entry failed (see errors)"
    # `value` is the FASTEST CPU path on the slab, all granted cores (a baseline must not be the
    # slower of two measured paths); both are named, with their figures -- or the same path on the whole domain
    cands = [c for c in (slab_f, slab_c) if c]
    slab = max(cands, key=lambda c: c["Mcells_s"]) if cands else None
    slab_sample = slab
    if full:
        slab = full
    impls = {"fortran": "fortran: oracle/fortran/advance_mu_t_cpu.f90, the build's own Fortran-90 restatement (fused, i blocks "
                        "marching in j, OpenMP j-tiles; amdflang -O3 -march=native -ffp-contract=off; bit-equal to the "
                        "reference's outputs in tests/golden/)",
             "port_c": "port_c: oracle/advance_mu_t_oracle_impl.h, the C restatement the parity tests check against (loop for loop "
                       "the reference's three phases; gcc -O3 -march=native -ffp-contract=off, OpenMP j-tiles)"}
    out = {"value": slab["Mcells_s"] if slab else None, "unit": "Mcells/s", "cores": slab["threads"] if slab else cores,
           "kind": "port",
           "impl": impls[slab["impl"]] if slab else impls["fortran"],
           "value_is": "the faster of the Fortran CPU path and the C port on the same j-slab and cores",
           "fortran_Mcells_s": slab_f["Mcells_s"] if slab_f else None,
           "port_c_Mcells_s": slab_c["Mcells_s"] if slab_c else None,
           "sample": (f"the WHOLE {ni}x{nk}x{nj} domain of the bench line ({domain_bytes / 2**30:.0f} GiB of host arrays), median sweep, "
                      f"{full['threads']} OpenMP j-tiles, pages first touched by their tile's thread" if full else
                      f"{ni}x{nk}x{slab_rows} j-slab (rows {gj0 + 1}..{gj0 + slab_rows}) of the same synthetic domain, median sweep, "
                      f"{slab['threads'] if slab else cores} OpenMP j-tiles, pages first touched by their tile's thread"),
           "sample_is": "full domain" if full else "j-slab",
           "full_domain_skipped_because": full_skipped,
           "slab_sample_Mcells_s": slab_sample["Mcells_s"] if slab_sample else None,
           "ms_per_sweep_sample": slab["ms_per_sweep"] if slab else None,
           "fastest_sweep_Mcells_s": slab.get("Mcells_s_fastest_sweep") if slab else None,   # the host is shared: its best sweep beside the median
           "one_thread_Mcells_s": one["Mcells_s"] if one else None,
           "host": quota_note + (f", MemAvailable {avail / 2**30:.0f} GiB" if avail else ""),
           "leg_seconds": round(time.perf_counter() - t_leg, 1),
           "build_seconds_not_in_the_budget": build_seconds,
           "matrix": matrix}
    ref = [m for m in matrix if m["impl"] == "reference_fortran_compute_only"]
    if ref:
        out["reference_fortran_compute_only_one_thread_Mcells_s"] = ref[0]["Mcells_s"]
        out["reference_note"] = ("reference_fortran_compute_only = /root/reference/module_small_step_em.f90 with lines 175-189 "
                                 "(five whole-array debug dumps, 99.6 % of its as-shipped wall time) cut out at build time, "
                                 "amdflang -O3 -ffp-contract=off, one thread (oracle/Makefile, target ref); "
                                 "reference_fortran_incl_dumps is the routine as shipped, dumps to /dev/null: an I/O number")
    if errors:
        out["errors"] = errors
    return out


def emulate_one_rank(a):
    """Per-GPU time of an N-slab run, measured on one GPU: the slab of one rank, the same three
    launches and two streams per sweep, the halo rows arriving by device-to-device copies instead
    of RCCL.  Projection only (no xGMI, no neighbour skew)."""
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    S = pkg.synth
    world = a.emulate_world
    rank = a.emulate_rank if a.emulate_rank >= 0 else world // 2
    torch.cuda.set_device(0)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    dims = (a.ni, a.nk, a.nj)
    gb = S.domain_bounds(*dims, aligned=True, align_elems=a.align_elems)
    sb = S.slab_bounds(gb, rank, world)
    dev = S.make_patch(sb, pkg.GridConfig(), dtype=dtype, seed=a.seed, global_dims=dims, device="cuda:0")
    src = {n: dev.arrays[n][-1].clone() for n in S.HALO_FROM_ABOVE}
    src_below = dev.arrays["t_1"][0].clone()

    def transport(st):
        if st.above is not None:
            for n in S.HALO_FROM_ABOVE:
                st.patch.arrays[n][-1].copy_(src[n], non_blocking=True)
        if st.below is not None:
            st.patch.arrays["t_1"][0].copy_(src_below, non_blocking=True)

    st = pkg.patch.SlabStepper(dev, rank, world, pkg.advance_mu_t, overlap=not a.no_overlap,
                               variant=a.variant, transport=transport)
    for _ in range(max(a.warmup, 1)):
        st.step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        st.step()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = wall * 1e3 / a.steps
    cells = a.ni * a.nk * a.nj
    print(json.dumps({"projection": f"one rank of {world} (rank {rank}, rows {sb.jts}..{sb.jte}) on one GPU",
                      "ms_per_step": round(ms, 4), "event_ms_per_step": round(ev0.elapsed_time(ev1) / a.steps, 4),
                      "projected_Mcells_s_all_ranks": round(cells / (ms * 1e-3) / 1e6, 1),
                      "note": "halo rows by device-to-device copy; no xGMI transfer, no neighbour skew"}), flush=True)


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: this process (which never touches a
    GPU) starts N rank processes of this same script with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, lets rank 0 print the JSON line on the inherited stdout, and
    returns the worst exit code.  Every rank runs in its own session (process group): a rank that
    dies takes the others down after 10 s, and so do a timeout, SIGTERM / SIGINT to this process and
    any exception here -- by process group id, nothing is ever matched by name.  The ranks' stderr
    comes through tagged `[rank r]`; RCCL runs with NCCL_DEBUG=WARN unless the caller set it."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    nonce = f"{os.getpid()}-{time.time_ns()}"
    procs, pumps = [], []

    def teardown(sig=signal.SIGKILL):
        for p in procs:                          # the exact process groups started below, nothing else
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        print(f"bench.py: signal {signum}: ending the {len(procs)} rank processes", file=sys.stderr, flush=True)
        teardown()
        raise SystemExit(128 + signum)

    def pump(stream, tag):
        for line in iter(stream.readline, b""):
            sys.stderr.write(f"[rank {tag}] " + line.decode(errors="replace"))
            sys.stderr.flush()
        stream.close()

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                       LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       AMT_RENDEZVOUS_NONCE=nonce, AMT_BENCH_SELF_LAUNCHED="1")
            if env.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":     # unset, or the image's quiet default
                env["NCCL_DEBUG"] = "WARN"
            # the ranks run the script this process was started as (bench.py; or the test shim that injects --cpu-dry-run's compute)
            p = subprocess.Popen([sys.executable, str(Path(sys.argv[0]).resolve())] + sys.argv[1:], env=env,
                                 stdout=None if r == 0 else subprocess.PIPE, stderr=subprocess.PIPE,
                                 start_new_session=True)
            procs.append(p)
            for stream in ((p.stderr,) if r == 0 else (p.stderr, p.stdout)):
                t = threading.Thread(target=pump, args=(stream, r), daemon=True)
                t.start()
                pumps.append(t)
        deadline = time.monotonic() + a.launch_timeout
        failed_at = None
        while any(p.poll() is None for p in procs):
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad and failed_at is None:
                failed_at = time.monotonic()
                rc = bad[0][1]
                print(f"bench.py: rank {bad[0][0]} exited with code {bad[0][1]}; the others get 10 s", file=sys.stderr, flush=True)
            timed_out = time.monotonic() > deadline
            if timed_out or (failed_at is not None and time.monotonic() - failed_at > 10.0):
                if timed_out:
                    alive = [r for r, p in enumerate(procs) if p.poll() is None]
                    print(f"bench.py: self-launch timed out after {a.launch_timeout:.0f} s; ranks still running: {alive}",
                          file=sys.stderr, flush=True)
                    rc = rc or 124
                teardown()
                break
            time.sleep(0.05)
        for p in procs:
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                pass
            if p.returncode and not rc:
                rc = p.returncode
    finally:
        teardown()
        for t in pumps:
            t.join(timeout=2)
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


class StepTimeout(RuntimeError):
    pass


# --cpu-dry-run: callable(*the 48 advance_mu_t arguments) that updates a tile of CPU tensors in place.  There is no CPU
# implementation in the product: tests/workers/bench_cpu_shim.py sets this to the oracle (test infrastructure) before main().
DRY_RUN_COMPUTE = None


def dry_run_rank(a):
    """One rank of `python <shim> --gpus N --cpu-dry-run`: everything of the N > 1 path that is not the GPU -- the
    self-launcher's environment, the gloo group, slab_bounds, the poisoned halos and the torch.distributed exchange of
    patch.SlabStepper, the first-sweep verification, barriers, max over ranks, the JSON line, the teardown."""
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    if DRY_RUN_COMPUTE is None:
        raise SystemExit("bench.py --cpu-dry-run: no compute callable injected (bench.DRY_RUN_COMPUTE); the product has no CPU path")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("AMT_BENCH_TEST_DIE_RANK") == str(rank):     # tests/test_bench_dry_run.py: a rank that dies before the group forms
        print(f"rank {rank} exiting with code 7 for the teardown test", file=sys.stderr, flush=True)
        os._exit(7)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(60.0, a.comm_timeout)))
    pkg = g.load_package()
    S = pkg.synth
    dtype = np.float64 if a.dtype == "f64" else np.float32
    dims = (a.ni, a.nk, a.nj)
    gb = S.domain_bounds(*dims, aligned=True, align_elems=a.align_elems)
    sb = S.slab_bounds(gb, rank, world)
    host = S.make_patch(sb, pkg.GridConfig(), dtype=dtype, seed=a.seed, global_dims=dims)
    arrays = {k: torch.from_numpy(v) for k, v in host.arrays.items()}
    if rank < world - 1:
        for name in S.HALO_FROM_ABOVE:
            arrays[name][-1].fill_(float("nan"))
    if rank > 0:
        arrays["t_1"][0].fill_(float("nan"))
    patch = S.Patch(sb, host.config, arrays, host.rdx, host.rdy, host.dts, host.epssm, dims)
    stepper = pkg.patch.SlabStepper(patch, rank, world, DRY_RUN_COMPUTE)
    dist.barrier()
    stepper.step()
    oracle = g.load_oracle()                                     # the checker, as in the GPU path
    verified, why = verify_first_sweep(pkg, oracle, patch, gb, dims, dtype, a.seed, (sb.jts, sb.jte))
    for _ in range(max(a.warmup - 1, 0)):
        stepper.step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        stepper.step()
    dist.barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall, 1.0 if verified else 0.0], dtype=torch.float64)
    tmax, tmin = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"dry_run": True, "note": "CPU plumbing run of the N > 1 path with an injected compute callable: not a measurement",
                          "n_gpus": world, "ranks_seen": dist.get_world_size(), "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(float(tmax[0]) * 1e3 / max(a.steps, 1), 4), "verified_vs_oracle": bool(tmin[1] > 0.5),
                          "verify_message": why or None, "scaling": "strong",
                          "config": {"workload": f"advance_mu_t {a.ni}x{a.nk}x{a.nj} {a.dtype}, {world} j-slab(s)",
                                     "rows_per_rank": [S.slab_bounds(gb, r, world).jte - S.slab_bounds(gb, r, world).jts + 1 for r in range(world)],
                                     "halo_transport": "gloo (CPU tensors)", "halo_bytes_per_rank_per_sweep": stepper.halo_bytes_per_sweep()},
                          "launched_by": "bench.py self-launch" if os.environ.get("AMT_BENCH_SELF_LAUNCHED") else "external launcher"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not bool(tmin[1] > 0.5):
        raise SystemExit(3)


def watchdog(fn, seconds, what):
    """Run fn() on a helper thread and wait at most `seconds`: the calls that can block for ever inside RCCL
    (ncclCommInitRank while a peer never arrives, the first send/recv of a connection) must not take the
    rank -- and with it the whole launch -- past the driver's own time limit without a diagnosis."""
    import threading
    box = {}

    def body():
        try:
            box["value"] = fn()
        except BaseException as e:  # noqa: BLE001
            box["error"] = e
    th = threading.Thread(target=body, daemon=True, name=f"amt-{what}")
    th.start()
    th.join(seconds)
    if th.is_alive():
        raise StepTimeout(f"{what}: no return after {seconds:.0f} s")
    if "error" in box:
        raise box["error"]
    return box.get("value")


def die(rank, code, msg):
    """End THIS rank now with a message (the self-launcher, or torchrun, then ends its peers): a rank that
    cannot go on must not sit in a collective its peers will never reach."""
    print(f"bench.py rank {rank}: FATAL: {msg}", file=sys.stderr, flush=True)
    os._exit(code)                               # not SystemExit: a helper thread may still be blocked inside RCCL


def run_rank(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("AMT_BENCH_TEST_HANG"):      # tests/test_bench_contract.py: a rank that never comes back
        print(f"rank {rank} pid {os.getpid()} hanging for the teardown test", file=sys.stderr, flush=True)
        time.sleep(float(os.environ["AMT_BENCH_TEST_HANG"]))
        raise SystemExit(0)
    smi_idle = gpu_state_smi() if rank == 0 and not a.no_box_probe else None     # before anything touches the GPU
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if world > 1 and a.transport == "ipc":
        os.environ["AMT_SLAB_TRANSPORT"] = "ipc"       # also makes amt_comm_unique_id independent of RCCL
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    if a.backend == "gloo":
        a.stepper = "torch"                        # bring-up mode: host-staged rows, ranks may share a GPU
        local_rank = local_rank % ndev
    elif world > 1 and a.share_gpu:
        local_rank = local_rank % ndev             # failure-path bring-up: RCCL will refuse two ranks on one device
    elif world > 1 and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible (RCCL needs one "
                         "GPU per rank; --backend gloo shares a GPU for bring-up)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    native = world > 1 and a.stepper == "native"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        pg_timeout = datetime.timedelta(seconds=max(60.0, 2 * a.comm_timeout + 60.0))     # never the 30-minute default
        if a.backend == "nccl" and not native:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=pg_timeout)
        else:
            # native stepper: RCCL lives inside the C++ runtime; torch.distributed (gloo, host side)
            # only carries the communicator id, the barriers and the max over ranks of the timings
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)

    pkg = g.load_package()
    S = pkg.synth
    if a.beside_rounds or a.beside_reserve:
        pkg.load_library().amt_march_set_beside(a.beside_rounds, a.beside_reserve)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    itemsize = np.dtype(dtype).itemsize
    dims = (a.ni, a.nk, a.nj)
    gb = S.domain_bounds(*dims, aligned=True, align_elems=a.align_elems)
    gb = gb.replace(ime=gb.ime + a.idim_extra)
    sb = S.slab_bounds(gb, rank, world)
    cfg = pkg.GridConfig()

    # every launch, copy and event of this rank goes to ONE stream (torch's current one)
    main_stream = torch.cuda.Stream(device=device) if native else torch.cuda.current_stream(device)
    torch.cuda.set_stream(main_stream)
    # The state is allocated by the PRODUCT: amt_domain_create, the call a Fortran or C host makes once, with its default
    # placement sampling (AMT_DOMAIN_PLACEMENT_TRIES allocations of the state, the fastest kept: the sweep time depends on which
    # physical pages the driver hands out, profiles/r05_placement.md).  The tensors below VIEW those arrays: `value` is what a
    # once-allocating host gets.  --probe-placements K overrides the number of tries (1 = first as it comes).
    if a.probe_placements > 0:
        os.environ["AMT_DOMAIN_PLACEMENT_TRIES"] = str(a.probe_placements)
    dev = S.make_patch(sb, cfg, dtype=dtype, seed=a.seed, global_dims=dims, device=device, native_domain=True)
    probe_ms = dev.owner.placement_ms() or None
    torch.cuda.synchronize()
    ceilings, clocks_under_load = None, None
    if not a.no_box_probe and torch.cuda.mem_get_info(device)[0] > (9 << 30):
        try:
            ceilings, clocks_under_load = box_ceilings(pkg, main_stream, device)
        except Exception as e:  # noqa: BLE001
            ceilings = {"error": f"{type(e).__name__}: {e}"}

    def poison_halos():
        # only a working exchange gives the right answer
        if world > 1:
            if rank < world - 1:
                for name in S.HALO_FROM_ABOVE:
                    dev.arrays[name][-1].fill_(float("nan"))
            if rank > 0:
                dev.arrays["t_1"][0].fill_(float("nan"))

    poison_halos()
    torch.cuda.synchronize()
    ranks_seen = 1
    native_error, p2p_group = None, None
    if native:
        # phase 1, no collective: every rank must be able to open RCCL; agree before anyone blocks
        # in ncclCommInitRank waiting for a rank that cannot come
        err = ""
        try:
            uid = [pkg.patch.NativeSlabStepper.comm_unique_id() if rank == 0 else None]
            if rank != 0:
                pkg.patch.NativeSlabStepper.comm_unique_id()
        except pkg.AmtError as e:
            err, uid = str(e), [None]
        flag = torch.tensor([0.0 if err else 1.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 0.5:
            raise SystemExit(f"rank {rank}: no communicator id for the native stepper ({a.transport}): {err or 'another rank failed'}")
        dist.broadcast_object_list(uid, src=0)
        # phase 2, collective (ncclCommInitRank).  It runs on a helper thread under a watchdog, so that every
        # rank reaches the agreement below within --comm-timeout whatever its peers do.  Outcomes, worst over ranks:
        #   every rank has its communicator            -> the native stepper is timed;
        #   clean failure (an error code) somewhere   -> all ranks go on with the torch.distributed stepper and
        #                                                 the line SAYS so (a cross-check measurement beats none);
        #   a rank still blocked inside RCCL           -> every rank ends itself, non-zero, with what it knows.
        def where():
            return (f"rank {rank}/{world} device {local_rank} MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')} "
                    f"NCCL_DEBUG={os.environ.get('NCCL_DEBUG', 'unset')} HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}")
        stepper, state = None, 2
        try:
            stepper = watchdog(lambda: pkg.patch.NativeSlabStepper(dev, rank, world, uid[0], stream=main_stream,
                                                                   overlap=not a.no_overlap, variant=a.variant,
                                                                   transport=a.transport),
                               a.comm_timeout, "amt_slab_create (ncclCommInitRank)" if a.transport == "rccl" else "amt_slab_create (IPC set-up)")
        except StepTimeout as e:
            native_error, state = f"{e}; {where()}", 0
        except pkg.AmtError as e:
            native_error, state = f"{e}; {where()}", 1
        flag = torch.tensor([float(state)], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 0.5:
            die(rank, 5, native_error or "another rank is blocked in ncclCommInitRank; ending this rank too")
        if flag.item() < 1.5:
            if stepper is not None:
                try:
                    watchdog(stepper.close, 20.0, "amt_slab_destroy")
                except Exception:  # noqa: BLE001
                    pass
            native, stepper = False, None
            native_error = native_error or "the native stepper could not be created on another rank"
            print(f"bench.py rank {rank}: native stepper unavailable ({native_error}); falling back to --stepper torch",
                  file=sys.stderr, flush=True)
            import datetime
            try:
                p2p_group = watchdog(lambda: dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=a.comm_timeout)),
                                     a.comm_timeout + 10, "torch.distributed nccl group for the fallback stepper")
            except Exception as e:  # noqa: BLE001
                die(rank, 5, f"no RCCL path at all: native: {native_error}; torch.distributed: {type(e).__name__}: {e}")
        else:
            ranks_seen = stepper.comm_info()[1]
    if not native:
        stepper = pkg.patch.SlabStepper(dev, rank, world, pkg.advance_mu_t, overlap=not a.no_overlap, group=p2p_group,
                                        variant=a.variant, stage_through_host=(a.backend == "gloo"))
        if world > 1:
            ranks_seen = dist.get_world_size()
    torch.cuda.synchronize()
    if world > 1:
        # establish the RCCL point-to-point connections outside any timed or verified step (the
        # first send/recv between two ranks builds their channels, which takes seconds);
        # the inputs are static, so an extra exchange changes nothing
        def first_exchange():
            torch.cuda.set_device(device)              # the helper thread has its own current device
            stepper.exchange_halos()
            torch.cuda.synchronize(device)
        try:
            watchdog(first_exchange, a.comm_timeout, "first halo exchange (RCCL connection set-up)")
        except Exception as e:  # noqa: BLE001  (a rank that cannot exchange must not leave its peers waiting for ever)
            die(rank, 6, f"{type(e).__name__}: {e}; rank {rank}/{world} device {local_rank}, stepper "
                         f"{'native' if native else 'torch'}, neighbours {[r for r in (rank - 1, rank + 1) if 0 <= r < world]}")
        poison_halos()                             # the verification must see the in-step exchange
        torch.cuda.synchronize()
        dist.barrier()

    verified, why = None, ""
    fp32_err = None
    oracle = None
    warm_done = 0
    if a.warmup > 0 and not a.no_verify:
        oracle = g.load_oracle()
        stepper.step()
        warm_done = 1
        torch.cuda.synchronize()
        verified, why = verify_first_sweep(pkg, oracle, dev, gb, dims, dtype, a.seed, (sb.jts, sb.jte))
        if a.dtype == "f32" and rank == 0:
            fp32_err = fp32_error_vs_fp64(pkg, oracle, dev, gb, dims, a.seed, (sb.jts, sb.jte))
    clock_ramp = None
    if not a.no_box_probe:
        try:
            clock_ramp = ramp_clocks(pkg, main_stream, dev.arrays["u"])
        except Exception as e:  # noqa: BLE001
            clock_ramp = {"error": f"{type(e).__name__}: {e}"}
    for _ in range(a.warmup - warm_done):
        stepper.step()

    def fence():
        torch.cuda.synchronize()
        if native:
            stepper.sync()                         # also reports a device-side wait for a neighbour that gave up (IPC transport)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]   # one per sweep (median)
    fence()
    t0 = time.perf_counter()
    ev0.record(main_stream)
    for k in range(a.steps):
        stepper.step()
        marks[k].record(main_stream)
    ev1.record(main_stream)
    fence()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    per_sweep = [(ev0 if k == 0 else marks[k - 1]).elapsed_time(marks[k]) for k in range(a.steps)]

    rank_ms = [ev_ms / max(a.steps, 1)] * 2                       # min, max over ranks of the event ms per sweep
    if world > 1:
        rdev = device if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([wall, ev_ms, 1.0 if verified in (None, True) else 0.0], device=rdev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = t.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_ms = [float(tmin[1]) / max(a.steps, 1), float(tmax[1]) / max(a.steps, 1)]
        wall, ev_ms = float(tmax[0]), float(tmax[1])
        if verified is not None:
            verified = bool(tmin[2] > 0.5)

    if rank == 0:
        cells = a.ni * a.nk * a.nj
        ms_per_step = wall * 1e3 / max(a.steps, 1)
        value = cells * a.steps / wall / 1e6
        ev_per_step_s = ev_ms * 1e-3 / max(a.steps, 1)
        abytes = algorithmic_bytes(a.ni, a.nk, a.nj, itemsize)
        achieved = abytes / world / ev_per_step_s / 1e9              # GB/s per GPU (slowest rank)
        traffic, traffic_source = None, None
        # reads / writes of one launch for the box's mixed streaming ceiling: the PMC record when there is one,
        # the algorithmic split (8 reads + 3 writes per cell, 10 + 4 per column) otherwise
        rd_bytes = itemsize * a.ni * a.nj * (8 * a.nk + 10) / world
        wr_bytes = itemsize * a.ni * a.nj * (3 * a.nk + 4) / world
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():
            try:
                rec = json.loads(tf.read_text()).get(f"{a.ni}x{a.nk}x{a.nj}_{a.dtype}_n{world}")
                if rec:
                    traffic = rec.get("hbm_bytes_per_launch")
                    if rec.get("read_bytes") and rec.get("write_bytes"):
                        rd_bytes, wr_bytes = rec["read_bytes"], rec["write_bytes"]
                    traffic_source = (f"profiles/hbm_traffic.json <- {rec.get('source', '?')} (rocprofv3 --pmc passes of "
                                      f"this command, collected by profiles/collect.sh; a recorded measurement, "
                                      f"not re-measured in this run)")
            except Exception:
                traffic = None
        out = {
            "metric": "advance_mu_t grid-cells/sec (Mcells/s) + achieved HBM GB/s",
            "value": round(value, 2),
            "unit": "Mcells/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(float(np.median(per_sweep)), 4) if per_sweep else None,   # rank 0's sweeps
            "per_sweep_ms": [round(x, 3) for x in per_sweep] if len(per_sweep) <= 100 else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic (seeded closed-form WRF-shaped fields, include/amt_synth.h)",
            "config": {"workload": f"advance_mu_t {a.ni}x{a.nk}x{a.nj} (i,k,j) {a.dtype}, "
                                   f"{world} j-slab(s), one-row halo exchange per sweep",
                       "ni": a.ni, "nk": a.nk, "nj": a.nj, "variant": a.variant,
                       "halo_overlap": (not a.no_overlap) if world > 1 else None,
                       "halo_transport": ((stepper.transport() if native else "rccl") if a.backend == "nccl"
                                          else "gloo-host-staged (bring-up)") if world > 1 else None,
                       "ranks_share_a_device": bool(world > 1 and world > ndev),
                       "halo_pull": (stepper.pull_mode() or None) if native else None,
                       "halo_schedule": (("host-waited: post, interior on its own, pull + boundary rows behind it" if os.environ.get("AMT_IPC_HOST_WAIT", "1") != "0"
                                          else "device-waited: waiting kernel first, interior beside it") if native and a.transport == "ipc" and not a.no_overlap
                                         else ("exchange beside the interior (2 rounds unless --beside-rounds)" if not a.no_overlap else "no overlap")) if world > 1 else None,
                       "interior_plan": {"beside_rounds": a.beside_rounds or 2, "beside_reserve_cus": a.beside_reserve} if world > 1 else None,
                       "halo_bytes_per_rank_per_sweep": stepper.halo_bytes_per_sweep(),
                       "placement_probe_ms": probe_ms,
                       "kernel": pkg.load_library().amt_march_last_kernel().decode()},
            "stepper": ((f"native amt_slab_* (C++ runtime, {'ncclSend/ncclRecv' if a.transport == 'rccl' else 'IPC peer copies + mailbox'})") if native else
                        "torch.distributed P2P (patch.SlabStepper)") if world > 1 else "single launch per sweep",
            "ranks_seen": ranks_seen,
            "native_stepper_error": native_error,
            "rank_ms_per_step_min_max": [round(x, 4) for x in rank_ms],
            "launched_by": "bench.py self-launch" if os.environ.get("AMT_BENCH_SELF_LAUNCHED") else
                           ("external launcher" if world > 1 else "direct"),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": abytes // world,
                         "kernel_ms_per_launch": round(ev_per_step_s * 1e3, 4),
                         "aggregate_GBps": round(abytes / ev_per_step_s / 1e9, 1)},
            "verified_vs_oracle": verified,
        }
        if world > 1 and native and a.transport == "rccl":
            out["faster_alternative"] = ("--transport ipc: the RCCL-free transport with its host-waited schedule (one rank of 8 in loopback: bare "
                                         "launch + 2.3 %, flat to 1.5 ms of neighbour lateness; RCCL + 6 % and half the lateness), profiles/r05_slab_ab.md")
        if world > 1 and world > ndev:
            out["note"] = (f"{world} ranks share {ndev} device(s): a correctness run of the whole N > 1 path (slabs, halo exchange, "
                           "verification, reductions); `value` is NOT a scaling measurement")
        if probe_ms and len(probe_ms) >= 2:
            # The K probes are the same sweeps on K allocations of the same state, made and chosen by amt_domain_create itself:
            # `value` is timed on the set it kept, i.e. on what ANY host of the library gets from one amt_domain_create.
            pm, pmin, p0 = float(np.median(probe_ms)), float(min(probe_ms)), float(probe_ms[0])
            out["placement"] = {
                "selection": f"amt_domain_create: fastest of {len(probe_ms)} allocations of the state (the library's default sampling; "
                             "AMT_DOMAIN_PLACEMENT_TRIES=1 or --probe-placements 1 = the first as it comes)",
                "probe_ms": probe_ms,
                "ms_per_step_placement_median": round(ev_per_step_s * 1e3 * pm / pmin, 4),
                "frac_placement_median": round(achieved / HBM_PEAK_GBS * pmin / pm, 4),
                "value_placement_median": round(value * pmin / pm, 2),
                "ms_per_step_first_allocation": round(ev_per_step_s * 1e3 * p0 / pmin, 4),
                "frac_first_allocation": round(achieved / HBM_PEAK_GBS * pmin / p0, 4),
                "note": "value / roofline.frac are what a host that calls amt_domain_create once gets (the sampling is inside the "
                        "product); the *_first_allocation figures are what hipMalloc as it comes would have given"}
        if ceilings and "box_copy_GBps" in ceilings:
            # Attribution of the sweep time to the box or to the kernel.  A copy moves one byte out per byte in;
            # this sweep reads 2.7x what it writes, and reads stream faster than writes, so the box's ceiling for
            # THIS mix is  reads / read_rate + writes / write_rate,  with the write rate inferred from the copy.
            cp, rdr = ceilings["box_copy_GBps"], ceilings["box_read_GBps"]
            inv_w = 2.0 / cp - 1.0 / rdr                                   # seconds per GB written
            mixed_ms = (rd_bytes / rdr + wr_bytes * max(inv_w, 1.0 / rdr)) / 1e6
            moved = traffic or (rd_bytes + wr_bytes)              # N > 1 records are per rank and sweep already
            out["roofline"].update({
                "box_copy_GBps": cp, "box_read_GBps": rdr,
                "box_write_GBps_inferred": round(1.0 / max(inv_w, 1.0 / rdr), 1),
                "box_mixed_ceiling_ms": round(mixed_ms, 4),
                "frac_of_box_copy": round(moved / ev_per_step_s / 1e9 / cp, 4),
                "frac_of_box_mixed": round(mixed_ms / (ev_per_step_s * 1e3), 4),
                "box_note": "box_* are this box's own streaming rates measured in this process just before the timed sweeps "
                            "(amt_calib_stream_rate, 2 x 4 GiB, 16 B per lane; the guide's figure for a good box is ~6.3 TB/s): "
                            "frac_of_box_copy = HBM bytes moved per launch / launch time / box_copy; frac_of_box_mixed = the time the "
                            "box needs to stream this launch's reads and writes at its own rates / launch time.  A low `frac` with "
                            "frac_of_box_mixed near 1 is a slow box, not a slow kernel."})
        elif ceilings:
            out["roofline"]["box_error"] = ceilings.get("error")
        if smi_idle is not None or clocks_under_load is not None:
            out["gpu_state"] = {"idle_before_run": smi_idle, "under_load_sysfs": clocks_under_load,
                                "clock_ramp_before_warmup": clock_ramp}
        if why:
            out["verify_message"] = why
        if fp32_err is not None:
            out["fp32_vs_fp64_oracle"] = {"max_abs_err_over_field_scale": float(f"{fp32_err:.3e}"),
                                          "stated_tolerance": 2e-5,
                                          "within_tolerance": bool(fp32_err <= 2e-5)}
        if world == 1 and not a.no_traffic:
            # same-run, same-box HBM traffic: this process gives its arrays back first
            rf = out["roofline"]
            try:
                del stepper, dev
                torch.cuda.empty_cache()
                rd_m, wr_m, note = measure_traffic(a)
                rf["traffic_recorded"], rf["traffic_recorded_source"] = rf.get("traffic"), rf.get("traffic_source")
                rf["traffic"], rf["traffic_source"] = int(rd_m + wr_m), note
                rf["traffic_read_bytes"], rf["traffic_write_bytes"] = int(rd_m), int(wr_m)
                rf["traffic_over_algorithmic"] = round((rd_m + wr_m) / abytes, 4)
                if ceilings and "box_copy_GBps" in ceilings:
                    cp, rdr = ceilings["box_copy_GBps"], ceilings["box_read_GBps"]
                    inv_w = max(2.0 / cp - 1.0 / rdr, 1.0 / rdr)
                    mixed_ms = (rd_m / rdr + wr_m * inv_w) / 1e6
                    rf["box_mixed_ceiling_ms"] = round(mixed_ms, 4)
                    rf["frac_of_box_copy"] = round((rd_m + wr_m) / ev_per_step_s / 1e9 / cp, 4)
                    rf["frac_of_box_mixed"] = round(mixed_ms / (ev_per_step_s * 1e3), 4)
            except Exception as e:  # noqa: BLE001  (the recorded value stays, and the line says why)
                rf["traffic_same_run_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(dims, a.dtype, a.seed, a.cpu_rows, a.cpu_seconds)
            except Exception as e:  # noqa: BLE001  (the ancillary leg never costs the GPU line)
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        if native:
            stepper.close()
        dist.destroy_process_group()
    if verified is False:
        raise SystemExit(3)


def main():
    a = parse()
    if a.emulate_world > 1:
        return emulate_one_rank(a)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))
    if a.cpu_dry_run:
        return dry_run_rank(a)
    return run_rank(a)


if __name__ == "__main__":
    main()
