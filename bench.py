#!/usr/bin/env python3
"""bench.py -- advance_mu_t sweeps on N MI355X, one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts N rank processes itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one advance_mu_t sweep (one acoustic sub-step's call) over the whole
4096 x 60 x 4096 fp64 domain (BASELINE.json configs[2]/[3]); inputs are resident in HBM before
the timed region.  With N > 1 the SAME domain is split into N j-slabs (strong scaling), each
rank trades its one-row input halos every sweep while its interior rows compute.

N > 1 is a FIRST-CONTACT LADDER (see the block comment above supervise()): the process the launcher starts per rank is a
supervisor that never touches a GPU; every transport is a rung run in fresh child processes under --rung-timeout -- rccl
(ncclSend/ncclRecv inside the C++ runtime amt_slab_*, the path a Fortran host calls), then ipc (peer copies + a shared-memory
mailbox between processes, which may share one device), torch.distributed P2P over RCCL only if neither ran.  --transport both
(default) times BOTH: `value` is RCCL's (north_star's transport) when its rung ran; `transports`, `ladder`, `preflight` say the
rest.  Every rung starts from NaN-poisoned halo rows, verifies its first sweep against the oracle and a later sweep after NEW
values in the exchanged fields and re-poisoned halos (an exchange that delivers once does not pass).  --stepper torch /
--backend gloo are the in-process bring-up modes.

Output keys beyond the driver's contract:
  roofline      algorithmic HBM bytes of one sweep (W*NI*NJ*(11*NK+14))
                divided by the HIP-event time of the kernel launches, against 8 TB/s per GPU
  cpu_baseline  the fastest CPU path -- the build's Fortran-90 restatement or the C port, both j-tiled over the
                host cores -- timed on the WHOLE domain where host memory and the leg's budget allow, else on a bounded
                j-slab sample of it (rank 0, N=1; the record says which)
  config.idim ... aligned, wrf_rows   (N = 1) the memory layout of the timed state (rows padded to whole 128-byte lines), and the same
                sweeps + PMC passes on WRF's own unpadded extents ims:ime = 0:NI+1 in the same run
  transports, ladder, preflight   (N > 1) every rung's outcome; per-rank roofline figures against 8 TB/s x N
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
    ap.add_argument("--transport", choices=("both", "rccl", "ipc"), default="both",
                    help="N > 1, native stepper: what carries the halo rows -- rccl = ncclSend/ncclRecv (north_star's transport); "
                         "ipc = hipIpcMemHandles + a shared-memory mailbox + copy-engine pulls between the processes of one "
                         "node (no RCCL; no compute unit held while the wire is busy; ranks may share a device); both (default) = "
                         "time one after the other, each in fresh rank processes: `value` is RCCL's (north_star's transport) when it "
                         "ran, the IPC figures stand beside it under `transports`")
    ap.add_argument("--rung-timeout", type=float, default=180.0,
                    help="N > 1: most seconds one rung of the ladder (one transport's child process of a rank: set-up, verification, "
                         "warm-up, timed sweeps) may take before the rank's supervisor ends it and goes on to the next rung")
    ap.add_argument("--rung-child", default="", help=argparse.SUPPRESS)     # internal: this process is one rank's child of a rung
    ap.add_argument("--wrf-rows-steps", type=int, default=5,
                    help="N = 1: after the headline (padded rows) re-create the state with WRF's own unpadded memory extents "
                         "(ims:ime = 0:NI+1) and time this many sweeps -> `wrf_rows` in the line; 0 = skip")
    ap.add_argument("--traffic-layouts", default="", help=argparse.SUPPRESS)  # internal: PMC child runs these --align-elems values in turn
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


def measure_traffic(a, layouts, seconds=150.0):
    """HBM bytes of one launch ON THIS BOX, in this run, for every memory layout in `layouts` (--align-elems values; 1 = WRF's own
    unpadded rows): two child runs of this same script under `rocprofv3 --kernel-trace --kernel-include-regex amt_ --pmc
    FETCH_SIZE` / `... WRITE_SIZE` -- separate passes, the program directly after `--`, as /opt/skills/guides/MI355X_MICROARCH.md
    prescribes -- each child building one layout after the other and running three sweeps on it; the launches are told apart by
    their dispatch order.  gfx950 correction of profiles/README.md: FETCH_SIZE counts half of a streamed read.  Returns
    ({align: (read_bytes, write_bytes)}, note) or raises; the caller has released its own arrays first.  Each pass is its own
    process group under a timeout."""
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
    per_layout = 3                                            # 1 warm-up + 2 sweeps per layout in the child
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="amt_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--output-format", "csv", "--kernel-trace", "--kernel-include-regex", "amt_march|amt_column",
               "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, str(Path(__file__).resolve()),
               "--ni", str(a.ni), "--nk", str(a.nk), "--nj", str(a.nj), "--dtype", a.dtype, "--variant", str(a.variant),
               "--seed", str(a.seed), "--idim-extra", str(a.idim_extra), "--traffic-layouts", ",".join(str(x) for x in layouts),
               "--steps", "2", "--warmup", "1"]
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
                for n, r in enumerate(csv.DictReader(fh)):
                    if r.get("Counter_Name") == counter and "amt_" in r.get("Kernel_Name", ""):
                        try:
                            order = int(r.get("Dispatch_Id") or n)
                        except ValueError:
                            order = n
                        vals.append((order, float(r["Counter_Value"])))
        shutil.rmtree(out, ignore_errors=True)
        vals = [v for _, v in sorted(vals)]
        if p.returncode != 0 or len(vals) != per_layout * len(layouts):
            raise RuntimeError(f"the {counter} pass gave {len(vals)} launches, expected {per_layout * len(layouts)} (exit {p.returncode})")
        for k, al in enumerate(layouts):
            mine = vals[k * per_layout: (k + 1) * per_layout]
            got.setdefault(al, {})[counter] = sum(mine) / len(mine)
    res = {al: (2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0) for al, c in got.items()}
    note = (f"measured in this run on this box: child runs of this command under rocprofv3 --kernel-trace --pmc FETCH_SIZE and "
            f"--pmc WRITE_SIZE (separate passes, {per_layout} launches per layout, layouts --align-elems {list(layouts)} one after the other "
            f"in each child); HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 calibration of profiles/README.md")
    return res, note


def traffic_child(a):
    """The program rocprofv3 runs for measure_traffic: for every layout of --traffic-layouts the state, 1 + 2 sweeps, nothing else."""
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    S = pkg.synth
    torch.cuda.set_device(0)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    os.environ["AMT_DOMAIN_PLACEMENT_TRIES"] = "1"
    for al in [int(x) for x in a.traffic_layouts.split(",")]:
        gb = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True, align_elems=al)
        gb = gb.replace(ime=gb.ime + a.idim_extra)
        dev = S.make_patch(gb, pkg.GridConfig(), dtype=dtype, seed=a.seed, global_dims=(a.ni, a.nk, a.nj), device="cuda:0", native_domain=True)
        call = pkg.advance_mu_t.bind(*dev.args(), variant=a.variant)
        for _ in range(a.warmup + a.steps):
            call()
        torch.cuda.synchronize()
        del call, dev
        torch.cuda.empty_cache()


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


def cpu_baseline(dims, dtype_name, seed, rows, seconds, prebuild=None):
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
    overlapped = prebuild is not None
    try:
        if prebuild is not None:                      # started at the top of the run, beside the GPU part: join it
            so, se = prebuild["proc"].communicate(timeout=600)
            rcode = prebuild["proc"].returncode
        else:
            r = subprocess.run([sys.executable, worker, "--prebuild"], capture_output=True, text=True, timeout=600)
            so, se, rcode = r.stdout, r.stderr, r.returncode
        built = json.loads(so.strip().splitlines()[-1]) if rcode == 0 and so.strip() else None
        if rcode != 0:
            errors.append(f"prebuild: exit {rcode}: {se.strip()[-300:]}")
        elif built and built.get("failed"):
            errors.extend(f"prebuild: {x}" for x in built["failed"])
    except Exception as e:  # noqa: BLE001
        errors.append(f"prebuild: {type(e).__name__}: {str(e)[-300:]}")
    build_seconds = round(time.perf_counter() - t_build, 1)          # overlapped: only what was left to wait for
    build_total = round(time.perf_counter() - prebuild["t0"], 1) if overlapped else build_seconds
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
           "build_overlapped_with_the_gpu_part": overlapped, "build_seconds_since_the_start_of_the_run": build_total,
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
    import shutil
    import tempfile
    launch_dir = tempfile.mkdtemp(prefix="amt_bench_")       # the rungs' rendezvous files: removed below however the ranks end
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
                       AMT_RENDEZVOUS_NONCE=nonce, AMT_BENCH_SELF_LAUNCHED="1", AMT_BENCH_LAUNCH_DIR=launch_dir)
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
        shutil.rmtree(launch_dir, ignore_errors=True)
    return rc


class StepTimeout(RuntimeError):
    pass


# --cpu-dry-run: callable(*the 48 advance_mu_t arguments) that updates a tile of CPU tensors in place.  There is no CPU
# implementation in the product: tests/workers/bench_cpu_shim.py sets this to the oracle (test infrastructure) before main().
DRY_RUN_COMPUTE = None


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


def start_cpu_prebuild():
    """The -march=native CPU libraries of THIS machine (oracle/_native/<cpu>/: the Fortran CPU path in both precisions, the two
    harnesses) are compiled by a child of their own, started before anything else of the run and joined in front of the CPU leg:
    on a cold box that is 20-45 s of amdflang / amdclang / gcc, which now runs BESIDE the GPU part of the line (compilers on a few
    host cores; the timed region is 20 launches and their events) instead of in front of the CPU leg, and costs nothing where the
    directory is already there."""
    import subprocess
    worker = str(ROOT / "oracle" / "cpu_bench.py")
    try:
        p = subprocess.Popen([sys.executable, worker, "--prebuild"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except OSError:
        return None
    return {"proc": p, "t0": time.perf_counter()}


def run_rank(a):
    """N = 1 (the driver's headline line), and the in-process bring-up modes of N > 1 (--backend gloo: halo rows staged through the
    host, ranks may share a GPU; --stepper torch: torch.distributed P2P ops over RCCL).  The default N > 1 path is supervise()."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("AMT_BENCH_TEST_HANG"):      # tests/test_bench_contract.py: a rank that never comes back
        print(f"rank {rank} pid {os.getpid()} hanging for the teardown test", file=sys.stderr, flush=True)
        time.sleep(float(os.environ["AMT_BENCH_TEST_HANG"]))
        raise SystemExit(0)
    if a.traffic_layouts:
        return traffic_child(a)
    prebuild = start_cpu_prebuild() if world == 1 and not a.no_cpu_baseline else None
    smi_idle = gpu_state_smi() if rank == 0 and not a.no_box_probe else None     # before anything touches the GPU
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    if a.backend == "gloo":
        local_rank = local_rank % ndev             # bring-up mode: host-staged rows, ranks may share a GPU
    elif world > 1 and a.share_gpu:
        local_rank = local_rank % ndev
    elif world > 1 and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible (RCCL needs one "
                         "GPU per rank; --backend gloo shares a GPU for bring-up)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        pg_timeout = datetime.timedelta(seconds=max(60.0, 2 * a.comm_timeout + 60.0))     # never the 30-minute default
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=pg_timeout)
        else:
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
    sides = S.neighbour_sides(0, rank, 1, world)

    # every launch, copy and event of this rank goes to ONE stream (torch's current one)
    main_stream = torch.cuda.current_stream(device)
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

    S.poison_halos(dev, sides)                 # only a working exchange gives the right answer (nothing for a world of one)
    torch.cuda.synchronize()
    stepper = pkg.patch.SlabStepper(dev, rank, world, pkg.advance_mu_t, overlap=not a.no_overlap,
                                    variant=a.variant, stage_through_host=(a.backend == "gloo"))
    ranks_seen = dist.get_world_size() if world > 1 else 1
    torch.cuda.synchronize()
    if world > 1:
        # establish the RCCL point-to-point connections outside any timed or verified step (the
        # first send/recv between two ranks builds their channels, which takes seconds)
        def first_exchange():
            torch.cuda.set_device(device)              # the helper thread has its own current device
            stepper.exchange_halos()
            torch.cuda.synchronize(device)
        try:
            watchdog(first_exchange, a.comm_timeout, "first halo exchange (RCCL connection set-up)")
        except Exception as e:  # noqa: BLE001  (a rank that cannot exchange must not leave its peers waiting for ever)
            die(rank, 6, f"{type(e).__name__}: {e}; rank {rank}/{world} device {local_rank}, torch stepper, "
                         f"neighbours {[r for r in (rank - 1, rank + 1) if 0 <= r < world]}")
        S.poison_halos(dev, sides)                 # the verification must see the in-step exchange
        torch.cuda.synchronize()
        dist.barrier()

    verified, why = None, ""
    verified_later = None
    fp32_err = None
    oracle = None
    warm_done = 0
    if a.warmup > 0 and not a.no_verify:
        oracle = g.load_oracle()
        stepper.step()
        warm_done = 1
        torch.cuda.synchronize()
        verified, why = verify_sweeps(pkg, oracle, dev, gb, dims, dtype, a.seed, (sb.jts, sb.jte), 1)
        if a.dtype == "f32" and rank == 0:
            fp32_err = fp32_error_vs_fp64(pkg, oracle, dev, gb, dims, a.seed, (sb.jts, sb.jte))
        if world > 1:
            # a second verified sweep on NEW values of the exchanged fields and re-poisoned halo rows: an exchange that delivers
            # once and never again passes the first check and fails this one
            S.refresh_exchanged_inputs(dev, a.seed, 1)
            S.poison_halos(dev, sides)
            stepper.step()
            warm_done = 2
            torch.cuda.synchronize()
            verified_later, why2 = verify_sweeps(pkg, oracle, dev, gb, dims, dtype, a.seed, (sb.jts, sb.jte), 2)
            why = why or why2
            verified = bool(verified and verified_later)
    clock_ramp = None
    if not a.no_box_probe:
        try:
            clock_ramp = ramp_clocks(pkg, main_stream, dev.arrays["u"])
        except Exception as e:  # noqa: BLE001
            clock_ramp = {"error": f"{type(e).__name__}: {e}"}
    for _ in range(max(a.warmup - warm_done, 0)):
        stepper.step()

    def fence():
        torch.cuda.synchronize()
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
        row_bytes = gb.idim * itemsize
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
                       # the memory layout of the timed state: WRF's (ims:ime, kms:kme, jms:jme), i fastest; `aligned` = the i extent is
                       # padded so that rows are whole 128-byte lines with i = its on a line boundary (a choice of ims:ime, which a WRF
                       # build makes).  WRF's own unpadded extents are timed beside it: `wrf_rows` below.
                       "idim": gb.idim, "kdim": gb.kdim, "jdim": gb.jdim, "ims": gb.ims, "ime": gb.ime,
                       "row_bytes": row_bytes, "row_bytes_mod_128": row_bytes % 128,
                       "aligned": bool(row_bytes % 128 == 0 and ((gb.its - gb.ims) * itemsize) % 128 == 0),
                       "halo_overlap": (not a.no_overlap) if world > 1 else None,
                       "halo_transport": ("rccl" if a.backend == "nccl" else "gloo-host-staged (bring-up)") if world > 1 else None,
                       "ranks_share_a_device": bool(world > 1 and world > ndev),
                       "halo_bytes_per_rank_per_sweep": stepper.halo_bytes_per_sweep(),
                       "placement_probe_ms": probe_ms,
                       "kernel": pkg.load_library().amt_march_last_kernel().decode()},
            "stepper": "torch.distributed P2P (patch.SlabStepper)" if world > 1 else "single launch per sweep",
            "ranks_seen": ranks_seen,
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
        if world > 1:
            out["verified_later_sweep_after_new_inputs"] = verified_later
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
        wrf = None
        if world == 1:
            del stepper, dev
            torch.cuda.empty_cache()
            # WRF's own unpadded memory extents (ims:ime = 0:NI+1: rows of NI+2 elements, 16 bytes past a line for 4098 x fp64), in the
            # same run on the same box: the state re-created through amt_domain_create like the headline's, 2 + N sweeps
            if a.wrf_rows_steps > 0 and a.align_elems != 1:
                try:
                    wrf = time_wrf_rows(a, pkg, device, dtype, abytes)
                except Exception as e:  # noqa: BLE001
                    wrf = {"error": f"{type(e).__name__}: {e}"}
                out["wrf_rows"] = wrf
        if world == 1 and not a.no_traffic:
            # same-run, same-box HBM traffic: this process has given its arrays back
            rf = out["roofline"]
            try:
                layouts = [a.align_elems] + ([1] if wrf and "error" not in wrf else [])
                res, note = measure_traffic(a, layouts)
                rd_m, wr_m = res[a.align_elems]
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
                if 1 in res and wrf and "error" not in wrf:
                    rd_w, wr_w = res[1]
                    wrf.update(traffic=int(rd_w + wr_w), traffic_read_bytes=int(rd_w), traffic_write_bytes=int(wr_w),
                               traffic_over_algorithmic=round((rd_w + wr_w) / abytes, 4))
            except Exception as e:  # noqa: BLE001  (the recorded value stays, and the line says why)
                rf["traffic_same_run_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(dims, a.dtype, a.seed, a.cpu_rows, a.cpu_seconds, prebuild=prebuild)
            except Exception as e:  # noqa: BLE001  (the ancillary leg never costs the GPU line)
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if verified is False:
        raise SystemExit(3)


def time_wrf_rows(a, pkg, device, dtype, abytes):
    """The same sweep on WRF's own memory extents -- ims:ime = 0:NI+1, no padding: rows of NI+2 elements -- on this box, in this
    run: state from amt_domain_create (placement sampling as for the headline), 2 warm-up + --wrf-rows-steps timed sweeps."""
    import torch
    S = pkg.synth
    gb = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True, align_elems=1)
    itemsize = np.dtype(dtype).itemsize
    dev = S.make_patch(gb, pkg.GridConfig(), dtype=dtype, seed=a.seed, global_dims=(a.ni, a.nk, a.nj), device=device, native_domain=True)
    probe = dev.owner.placement_ms() or None
    call = pkg.advance_mu_t.bind(*dev.args(), variant=a.variant)
    for _ in range(2):
        call()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.wrf_rows_steps + 1)]
    stream = torch.cuda.current_stream(device)
    ev[0].record(stream)
    for k in range(a.wrf_rows_steps):
        call()
        ev[k + 1].record(stream)
    torch.cuda.synchronize()
    per = [ev[k].elapsed_time(ev[k + 1]) for k in range(a.wrf_rows_steps)]
    ms = ev[0].elapsed_time(ev[-1]) / a.wrf_rows_steps
    kernel = pkg.load_library().amt_march_last_kernel().decode()
    del call, dev
    torch.cuda.empty_cache()
    return {"layout": "WRF's own memory extents ims:ime = 0:NI+1 (bench.py --align-elems 1): no padding",
            "idim": gb.idim, "row_bytes": gb.idim * itemsize, "row_bytes_mod_128": (gb.idim * itemsize) % 128,
            "steps": a.wrf_rows_steps, "ms_per_step": round(ms, 4), "ms_per_step_median": round(float(np.median(per)), 4),
            "frac": round(abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "achieved_GBps": round(abytes / (ms * 1e-3) / 1e9, 1),
            "placement_probe_ms": probe, "kernel": kernel}


# =====================================================================================================================
# N > 1: supervisors and rungs.
#
# The process a launcher starts per rank (bench.py's own self-launch, or torch.distributed.run) is a SUPERVISOR: it never
# touches a GPU.  The supervisors hold a gloo group among themselves (host side only: agreement, gathering the records) and
# walk a LADDER of rungs; for every rung each supervisor starts a FRESH child process of this script (`--rung-child <name>`)
# that sets its device, builds its slab, creates the stepper and measures -- and that the supervisor ends, by its exact pid,
# if it does not come back within --rung-timeout.  Whatever a transport does on first contact with real hardware (an
# ncclCommInitRank that never returns, a hipIpcOpenMemHandle that refuses, a kernel that waits for ever) costs one rung on
# every rank, never the launch, and no process that has initialised the GPU is ever re-used or re-exec'ed.  The children of a
# rung need no torch.distributed: the communicator id travels through the library's file rendezvous and the barriers / max
# over ranks through amt_slab_barrier / amt_slab_max -- exactly what a Fortran host without MPI uses.
#
#   preflight        device count, peer-access matrix, librccl, /dev/shm  (never fatal by itself: recorded)
#   rccl             native stepper, ncclSend/ncclRecv, device-waited schedule       } --transport both: both are timed
#   ipc              native stepper, IPC mailbox + pulls, host-waited schedule       } and verified, each on fresh processes
#   torch-rccl       torch.distributed P2P stepper over RCCL: only if no native rung succeeded
#
# `value` is north_star's transport (RCCL) when its rung succeeded, else the IPC rung's, else torch's; every rung's outcome
# is in the line (`ladder`, `transports`).  advance_mu_t_no_async.cu:329-357 is what this replaces: the reference launches and
# synchronises its devices from one host thread and has no failure path but exit(1).
# =====================================================================================================================
RUNG_FAIL_EXIT = 21          # a rung's child that failed cleanly (its JSON record says why)


def verify_sweeps(pkg, oracle, dev, gb, dims, dtype, seed, rank_rows, sweeps=1):
    """After exactly `sweeps` sweeps from fresh inputs -- sweep s > 0 having run on the exchanged inputs of seed + s
    (synth.refresh_exchanged_inputs) -- recompute a few 3-row j-slabs with the oracle from regenerated inputs, the same
    refills between its sweeps, and compare bit for bit (size-independent parity check; the slabs sit on this rank's first
    and last rows, where only a halo row that arrived in THIS sweep gives the right bits)."""
    S = pkg.synth
    b = dev.bounds
    jlo_own, jhi_own = rank_rows
    cand = sorted({jlo_own, max(jlo_own, jhi_own - 2), (jlo_own + jhi_own) // 2})
    for jlo in cand:
        jhi = min(jlo + 2, jhi_own)
        sb = gb.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)
        want = S.make_patch(sb, dev.config, dtype=dtype, seed=seed, global_dims=dims)
        for s in range(sweeps):
            if s:
                S.refresh_exchanged_inputs(want, seed, s)
            oracle.advance_mu_t_omp(*want.args(), nthreads=min(3, jhi - jlo + 1))
        for n in S.OUTPUTS:
            got = dev.arrays[n][jlo - b.jms: jhi + 1 - b.jms]
            got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
            w = want.arrays[n][1: 1 + (jhi - jlo + 1)]
            if not np.array_equal(np.ascontiguousarray(got).view(np.uint8), w.view(np.uint8)):
                return False, f"rows {jlo}..{jhi} of {n} differ from the oracle after sweep {sweeps}"
    return True, ""


RUNG_RECORD_MARK = "AMT_RUNG_RECORD "


def _emit(rec, code=0):
    """A rung child's one record: a line of its own behind a marker (libraries write to the same stdout -- RCCL's warnings do not
    always end their lines), which its supervisor looks for from the end; then exit."""
    sys.stdout.write("\n" + RUNG_RECORD_MARK + json.dumps(rec) + "\n")
    sys.stdout.flush()
    os._exit(code)                      # not SystemExit: a helper thread may still sit inside RCCL


def _injected_outcome(kind, rank):
    """tests/test_bench_dry_run.py: AMT_BENCH_TEST_RUNG_<KIND>=refuse|hang|open_fails[:rank] makes this rung's child (of that
    rank, default every rank) fail cleanly at set-up, never come back, or fail like a refused IPC handle."""
    spec = os.environ.get("AMT_BENCH_TEST_RUNG_" + kind.upper().replace("-", "_"), "")
    if not spec:
        return None
    what, _, who = spec.partition(":")
    if who and int(who) != rank:
        return None
    return what


def preflight_child(a):
    """What the node looks like before any transport is tried: devices, peer access, librccl, /dev/shm.  Touches the GPU
    (a fresh process per rank, under the supervisor's timeout)."""
    import ctypes
    rank = int(os.environ.get("RANK", "0"))
    rec = {"rung": "preflight", "rank": rank, "ok": True}
    if a.cpu_dry_run:
        rec.update(devices=0, dry_run=True)
        _emit(rec)
    import torch
    if not torch.cuda.is_available():
        rec.update(ok=False, error="bench.py needs a GPU (there is no CPU fallback for the product path)")
        print(rec["error"], file=sys.stderr, flush=True)
        _emit(rec, RUNG_FAIL_EXIT)
    ndev = torch.cuda.device_count()
    rec["devices"] = ndev
    rec["device_name"] = torch.cuda.get_device_name(0)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if local_rank >= ndev and not a.share_gpu:
        # fewer devices than ranks: say so NOW (every rung would otherwise wait --comm-timeout for the ranks that cannot come)
        rec.update(ok=False, error=f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible: one GPU per rank (--share-gpu maps the ranks onto "
                                   "the visible devices: a correctness run over the IPC transport, never a scaling number)")
        print(rec["error"], file=sys.stderr, flush=True)
        _emit(rec, RUNG_FAIL_EXIT)
    if rank == 0:
        try:
            rec["peer_access"] = [[1 if i == j else int(torch.cuda.can_device_access_peer(i, j)) for j in range(ndev)] for i in range(ndev)]
        except Exception as e:  # noqa: BLE001
            rec["peer_access"] = f"unavailable ({type(e).__name__}: {e})"
    try:
        lib = None
        for name in (os.environ.get("AMT_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
            if name:
                try:
                    lib = ctypes.CDLL(name)
                    break
                except OSError:
                    continue
        if lib is None:
            raise OSError("librccl not found")
        v = ctypes.c_int(0)
        lib.ncclGetVersion(ctypes.byref(v))
        rec["rccl"] = {"loadable": True, "version": v.value}
    except Exception as e:  # noqa: BLE001
        rec["rccl"] = {"loadable": False, "error": f"{type(e).__name__}: {e}"}
    try:
        st = os.statvfs("/dev/shm")
        rec["dev_shm_free_MiB"] = int(st.f_bavail * st.f_frsize / 2**20)
    except OSError as e:
        rec["dev_shm_free_MiB"] = f"unavailable ({e})"
    rec["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    _emit(rec)


def rung_child(a):
    """One rank's child of one rung: everything that touches the GPU (or, with --cpu-dry-run, the CPU stand-in with the compute
    callable the test shim injects).  Prints ONE JSON record and exits 0, or RUNG_FAIL_EXIT with the error in the record."""
    kind = a.rung_child
    if kind == "preflight":
        return preflight_child(a)
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    launch_dir = os.environ["AMT_BENCH_DIR"]
    rec = {"rung": kind, "rank": rank, "ok": False}
    injected = _injected_outcome(kind, rank)
    if injected == "hang":
        print(f"rank {rank} rung {kind} pid {os.getpid()}: hanging (injected)", file=sys.stderr, flush=True)
        time.sleep(3600)
    if injected in ("refuse", "open_fails"):
        rec["error"] = ("ncclCommInitRank failed: invalid usage (injected)" if injected == "refuse"
                        else "hipIpcOpenMemHandle of rank 1's rows failed: invalid argument (injected)")
        _emit(rec, RUNG_FAIL_EXIT)

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    S = pkg.synth
    dtype = np.float64 if a.dtype == "f64" else np.float32
    itemsize = np.dtype(dtype).itemsize
    dims = (a.ni, a.nk, a.nj)
    gb = S.domain_bounds(*dims, aligned=True, align_elems=a.align_elems)
    gb = gb.replace(ime=gb.ime + a.idim_extra)
    sb = S.slab_bounds(gb, rank, world)
    cfg = pkg.GridConfig()
    sides = S.neighbour_sides(0, rank, 1, world)
    dry = a.cpu_dry_run
    transport = {"rccl": "rccl", "ipc": "ipc"}.get(kind)
    dist = None

    def fail(msg, code=RUNG_FAIL_EXIT):
        rec["error"] = msg
        print(f"bench.py rank {rank} rung {kind}: {msg}", file=sys.stderr, flush=True)
        _emit(rec, code)

    # ---- the state and the stepper ------------------------------------------------------------------------------------
    native = transport is not None and not dry
    if dry:
        if DRY_RUN_COMPUTE is None:
            fail("bench.py --cpu-dry-run: no compute callable injected (bench.DRY_RUN_COMPUTE); the product has no CPU path", 2)
        import datetime
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method=f"file://{launch_dir}/store_{kind}", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=max(30.0, a.comm_timeout)))
        host = S.make_patch(sb, cfg, dtype=dtype, seed=a.seed, global_dims=dims)
        dev = S.Patch(sb, host.config, {k: torch.from_numpy(v) for k, v in host.arrays.items()}, host.rdx, host.rdy, host.dts, host.epssm, dims)
        S.poison_halos(dev, sides)
        stepper = pkg.patch.SlabStepper(dev, rank, world, DRY_RUN_COMPUTE)
        ndev, device, main_stream = 0, None, None
    else:
        if not torch.cuda.is_available():
            fail("bench.py needs a GPU (there is no CPU fallback for the product path)", 2)
        ndev = torch.cuda.device_count()
        if local_rank >= ndev and not a.share_gpu:
            fail(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible (one GPU per rank; --share-gpu --transport ipc shares them)", 2)
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        if transport == "ipc":
            os.environ["AMT_SLAB_TRANSPORT"] = "ipc"       # also makes the communicator id independent of RCCL
        if a.beside_rounds or a.beside_reserve:
            pkg.load_library().amt_march_set_beside(a.beside_rounds, a.beside_reserve)
        main_stream = torch.cuda.Stream(device=device) if native else torch.cuda.current_stream(device)
        torch.cuda.set_stream(main_stream)
        if a.probe_placements > 0:
            os.environ["AMT_DOMAIN_PLACEMENT_TRIES"] = str(a.probe_placements)
        dev = S.make_patch(sb, cfg, dtype=dtype, seed=a.seed, global_dims=dims, device=device, native_domain=True)
        rec["placement_probe_ms"] = dev.owner.placement_ms() or None
        S.poison_halos(dev, sides)                          # only a working exchange gives the right answer
        torch.cuda.synchronize()
        if native:
            import ctypes
            uid = (ctypes.c_char * 128)()
            try:
                pkg.lib.check(pkg.load_library().amt_comm_rendezvous_file(f"{launch_dir}/uid_{kind}".encode(), 0, rank, world,
                                                                          float(a.comm_timeout), uid))
                stepper = pkg.patch.NativeSlabStepper(dev, rank, world, bytes(uid), stream=main_stream, overlap=not a.no_overlap,
                                                      variant=a.variant, transport=transport)
            except pkg.AmtError as e:
                fail(f"{e}; rank {rank}/{world} device {local_rank} NCCL_DEBUG={os.environ.get('NCCL_DEBUG', 'unset')} "
                     f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}")
        else:                                               # torch-rccl: torch.distributed P2P ops around the same launches
            import datetime
            import torch.distributed as dist
            try:
                dist.init_process_group("nccl", init_method=f"file://{launch_dir}/store_{kind}", rank=rank, world_size=world,
                                        device_id=device, timeout=datetime.timedelta(seconds=max(60.0, a.comm_timeout)))
            except Exception as e:  # noqa: BLE001
                fail(f"torch.distributed nccl group: {type(e).__name__}: {e}")
            stepper = pkg.patch.SlabStepper(dev, rank, world, pkg.advance_mu_t, overlap=not a.no_overlap, variant=a.variant)

    def sync():
        if not dry:
            torch.cuda.synchronize()
        if native:
            stepper.sync()                 # also reports a device-side wait for a neighbour that gave up (IPC transport)

    def barrier():
        sync()
        if native:
            pkg.lib.check(stepper.L.amt_slab_barrier(stepper._slab))
        else:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def rank_max(x):
        if native:
            import ctypes
            v = ctypes.c_double(float(x))
            pkg.lib.check(stepper.L.amt_slab_max(stepper._slab, ctypes.byref(v)))
            return v.value
        t = torch.tensor([float(x)], dtype=torch.float64, device=device if (dist.get_backend() == "nccl") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def step(n=1):
        if native:
            stepper.step(n)
        else:
            for _ in range(n):
                stepper.step()

    def next_inputs(sweep):
        """new values in the fields that cross a slab boundary (the stand-in for advance_uv) and NaN in the halo rows again"""
        if native:
            stepper.next_substep_inputs(a.seed, sweep)
        else:
            S.refresh_exchanged_inputs(dev, a.seed, sweep)
            S.poison_halos(dev, sides)

    try:
        rec["ranks_seen"] = stepper.comm_info()[1] if native else dist.get_world_size()
        # the point-to-point connections are built by the first exchange, which takes seconds: outside anything timed or
        # verified (the inputs are what they are: an extra exchange changes nothing; the halos are poisoned again after it)
        stepper.exchange_halos()
        sync()
        S.poison_halos(dev, sides)
        barrier()
        # ---- verification: sweep 1 from poisoned halos, and sweep 2 after NEW inputs and re-poisoned halos (an exchange that
        # delivers once and never again passes the first check and fails the second)
        verified_first = verified_later = None
        why = ""
        warm_done = 0
        if a.warmup > 0 and not a.no_verify:
            oracle = g.load_oracle()
            step()
            sync()
            verified_first, why = verify_sweeps(pkg, oracle, dev, gb, dims, dtype, a.seed, (sb.jts, sb.jte), 1)
            next_inputs(1)
            step()
            sync()
            verified_later, why2 = verify_sweeps(pkg, oracle, dev, gb, dims, dtype, a.seed, (sb.jts, sb.jte), 2)
            why = why or why2
            warm_done = 2
            if a.dtype == "f32" and rank == 0 and not dry:
                rec["fp32_note"] = "fp32 vs the fp64 oracle is measured by the N = 1 line (fp32_vs_fp64_oracle)"
        for _ in range(max(a.warmup - warm_done, 0)):
            step()
        # ---- the timed sweeps: barrier + synchronize on both sides, max over ranks
        marks = []
        if not dry:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
        barrier()
        t0 = time.perf_counter()
        if not dry:
            ev0.record(main_stream)
        for k in range(a.steps):
            step()
            if not dry:
                marks[k].record(main_stream)
        if not dry:
            ev1.record(main_stream)
        barrier()
        wall = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1) if not dry else wall * 1e3
        per_sweep = [(ev0 if k == 0 else marks[k - 1]).elapsed_time(marks[k]) for k in range(a.steps)] if not dry else []
        rec.update(ok=True, wall_s=wall, wall_s_max_over_ranks=rank_max(wall), event_ms=ev_ms,
                   per_sweep_ms=[round(x, 3) for x in per_sweep] if len(per_sweep) <= 100 else None,
                   verified_first_sweep=verified_first, verified_later_sweep=verified_later, verify_message=why or None,
                   rows=[sb.jts, sb.jte], halo_bytes_per_sweep=stepper.halo_bytes_per_sweep(),
                   transport=(stepper.transport() if native else ("gloo (CPU tensors)" if dry else "rccl")),
                   pull=(stepper.pull_mode() or None) if native else None,
                   stepper=("native amt_slab_*" if native else "torch.distributed P2P (patch.SlabStepper)"),
                   kernel=None if dry else pkg.load_library().amt_march_last_kernel().decode(),
                   devices_visible=ndev, device=None if dry else local_rank)
        if verified_first is False or verified_later is False:
            rec["ok"] = False
            rec["error"] = f"results differ from the oracle: {why}"
        barrier()
    except Exception as e:  # noqa: BLE001  (AmtError(ERR_COMM) of a wait that gave up, an RCCL error, ...)
        fail(f"{type(e).__name__}: {e}")
    try:
        if native:
            stepper.close()
        elif dist is not None:
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    _emit(rec, 0 if rec["ok"] else RUNG_FAIL_EXIT)


def _set_pdeathsig():
    """preexec of a rung child: die with the supervisor (a launcher that kills the supervisor by pid must not leave a child
    with a GPU context behind)."""
    import ctypes
    import signal
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)       # PR_SET_PDEATHSIG
    except Exception:  # noqa: BLE001
        pass


def run_rung_child(a, kind, env, timeout):
    """Start this rank's child of rung `kind`, wait at most `timeout` seconds, return its record (the last JSON line of its
    stdout) -- or a record that says what happened instead.  The child is ended by its exact pid, never by a pattern."""
    import subprocess
    import threading
    rank = int(os.environ.get("RANK", "0"))
    argv = [sys.executable, str(Path(sys.argv[0]).resolve())] + [x for x in sys.argv[1:]] + ["--rung-child", kind]
    t0 = time.perf_counter()
    p = subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, preexec_fn=_set_pdeathsig)
    err_tail, out_chunks = [], []

    def pump_err():                          # the child's stderr goes on to this rank's stderr, tagged with the rung
        for line in iter(p.stderr.readline, b""):
            text = line.decode(errors="replace")
            err_tail.append(text)
            del err_tail[:-30]
            sys.stderr.write(f"[{kind}] " + text)
            sys.stderr.flush()

    def pump_out():                          # its stdout is kept: the record is in there
        for chunk in iter(lambda: p.stdout.read(65536), b""):
            out_chunks.append(chunk)
    pumps = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for th in pumps:
        th.start()
    timed_out = False
    try:
        p.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        p.kill()                             # this child, by its pid
        p.wait()
    for th in pumps:
        th.join(timeout=5)
    out = b"".join(out_chunks)
    seconds = round(time.perf_counter() - t0, 1)
    rec = None
    text = (out or b"").decode(errors="replace")
    at = text.rfind(RUNG_RECORD_MARK)
    if at >= 0:
        try:
            rec, _ = json.JSONDecoder().raw_decode(text[at + len(RUNG_RECORD_MARK):])
        except ValueError:
            rec = None
    if timed_out:
        rec = {"rung": kind, "rank": rank, "ok": False, "timed_out": True,
               "error": f"no return within {timeout:.0f} s (--rung-timeout): the child was ended by its pid; last stderr: "
                        + "".join(err_tail[-3:]).strip()[-300:]}
    elif rec is None:
        rec = {"rung": kind, "rank": rank, "ok": False,
               "error": f"the child exited with code {p.returncode} without a record; last stdout: {text.strip()[-200:]!r}; last stderr: "
                        + "".join(err_tail[-5:]).strip()[-400:]}
    elif p.returncode != 0 and rec.get("ok"):
        rec["ok"], rec["error"] = False, f"the child exited with code {p.returncode} after reporting success"
    rec["exit_code"], rec["seconds"] = p.returncode, seconds
    return rec


def supervise(a):
    """One rank's supervisor of an N > 1 run (see the block comment above)."""
    import datetime
    import shutil
    import tempfile
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if os.environ.get("AMT_BENCH_TEST_HANG"):      # tests/test_bench_contract.py: a rank that never comes back
        print(f"rank {rank} pid {os.getpid()} hanging for the teardown test", file=sys.stderr, flush=True)
        time.sleep(float(os.environ["AMT_BENCH_TEST_HANG"]))
        raise SystemExit(0)
    if os.environ.get("AMT_BENCH_TEST_DIE_RANK") == str(rank):     # tests/test_bench_dry_run.py: a rank that dies before the group forms
        print(f"rank {rank} exiting with code 7 for the teardown test", file=sys.stderr, flush=True)
        os._exit(7)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=max(120.0, 2 * a.rung_timeout + 60.0)))   # never the 30-minute default
    shared = [None]
    if rank == 0:
        given = os.environ.get("AMT_BENCH_LAUNCH_DIR")           # self_launch owns (and removes) the directory; under another launcher rank 0 does
        shared[0] = {"dir": given if given and os.path.isdir(given) else tempfile.mkdtemp(prefix="amt_bench_"),
                     "nonce": f"{os.getpid()}-{time.time_ns()}"}
        if shared[0]["dir"] != given:
            import atexit
            atexit.register(shutil.rmtree, shared[0]["dir"], ignore_errors=True)  # also when a rung or the group raises
    dist.broadcast_object_list(shared, src=0)
    launch_dir, nonce = shared[0]["dir"], shared[0]["nonce"]
    env = dict(os.environ, AMT_BENCH_DIR=launch_dir, AMT_RENDEZVOUS_NONCE=nonce)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)

    def rung(kind, timeout):
        """run this rank's child of the rung; every supervisor gets every rank's record"""
        mine = run_rung_child(a, kind, dict(env, AMT_RENDEZVOUS_NONCE=f"{nonce}-{kind}"), timeout)
        recs = [None] * world
        dist.all_gather_object(recs, mine)
        return recs

    ladder = []
    pre = rung("preflight", min(a.rung_timeout, 120.0))
    ladder.append({"rung": "preflight", "ok": all(r["ok"] for r in pre), "seconds": max(r["seconds"] for r in pre),
                   "errors": [f"rank {r['rank']}: {r['error']}" for r in pre if not r["ok"]] or None})
    if not all(r["ok"] for r in pre):
        # without a GPU (or with a rank that cannot even look at its node) there is nothing to measure
        if rank == 0:
            print("bench.py: preflight failed: " + "; ".join(ladder[0]["errors"]), file=sys.stderr, flush=True)
            shutil.rmtree(launch_dir, ignore_errors=True)
        dist.barrier()
        dist.destroy_process_group()
        raise SystemExit(2)
    kinds = {"both": ["rccl", "ipc"], "rccl": ["rccl"], "ipc": ["ipc"]}[a.transport]
    results = {}
    for kind in kinds:
        recs = rung(kind, a.rung_timeout)
        results[kind] = recs
        ok = all(r["ok"] for r in recs)
        ladder.append({"rung": kind, "ok": ok, "seconds": max(r["seconds"] for r in recs),
                       "errors": [f"rank {r['rank']}: {r.get('error')}" for r in recs if not r["ok"]] or None})
        if rank == 0 and not ok:
            print(f"bench.py: rung {kind} failed: " + "; ".join(ladder[-1]["errors"])[:1500], file=sys.stderr, flush=True)
    if not any(all(r["ok"] for r in results[k]) for k in kinds) and not a.cpu_dry_run and "rccl" in kinds:
        # no native rung: the torch.distributed stepper over RCCL as the last resort (a cross-check measurement beats none)
        recs = rung("torch-rccl", a.rung_timeout)
        results["torch-rccl"] = recs
        ladder.append({"rung": "torch-rccl", "ok": all(r["ok"] for r in recs), "seconds": max(r["seconds"] for r in recs),
                       "errors": [f"rank {r['rank']}: {r.get('error')}" for r in recs if not r["ok"]] or None})
    good = [k for k in results if all(r["ok"] for r in results[k])]
    rc = 0
    if rank == 0:
        line = supervisor_line(a, world, pre, results, ladder, good)
        print(json.dumps(line), flush=True)
        shutil.rmtree(launch_dir, ignore_errors=True)
    if not good:
        rc = 5
    elif any(r.get("verified_first_sweep") is False or r.get("verified_later_sweep") is False for k in results for r in results[k]):
        rc = 3
    dist.barrier()
    dist.destroy_process_group()
    if rc:
        raise SystemExit(rc)


def summarize_rung(a, world, recs):
    """One transport's figures from its ranks' records (max over ranks of the wall time of the K sweeps)."""
    cells = a.ni * a.nk * a.nj
    itemsize = 8 if a.dtype == "f64" else 4
    if not all(r["ok"] for r in recs):
        return {"ok": False, "errors": [f"rank {r['rank']}: {r.get('error')}" for r in recs if not r["ok"]],
                "timed_out_ranks": [r["rank"] for r in recs if r.get("timed_out")] or None}
    wall = max(max(r["wall_s"], r.get("wall_s_max_over_ranks") or 0.0) for r in recs)
    ev = [r["event_ms"] / max(a.steps, 1) for r in recs]
    abytes = algorithmic_bytes(a.ni, a.nk, a.nj, itemsize)
    rows = [r["rows"][1] - r["rows"][0] + 1 for r in recs]
    per_rank_gbps = [itemsize * a.ni * n * (11 * a.nk + 14) / (ms * 1e-3) / 1e9 for n, ms in zip(rows, ev)]
    verified = [r.get("verified_first_sweep") for r in recs] + [r.get("verified_later_sweep") for r in recs]
    return {"ok": True,
            "value": round(cells * a.steps / wall / 1e6, 2), "unit": "Mcells/s",
            "ms_per_step": round(wall * 1e3 / max(a.steps, 1), 4),
            "rank_event_ms_per_step_min_max": [round(min(ev), 4), round(max(ev), 4)],
            "verified_first_sweep": all(r.get("verified_first_sweep") is not False for r in recs) if any(v is not None for v in verified) else None,
            "verified_later_sweep_after_new_inputs": all(r.get("verified_later_sweep") is not False for r in recs) if any(v is not None for v in verified) else None,
            "stepper": recs[0]["stepper"], "halo_transport": recs[0]["transport"], "halo_pull": recs[0].get("pull"),
            "ranks_seen": min(r.get("ranks_seen") or 0 for r in recs),
            "halo_bytes_per_rank_per_sweep": [r["halo_bytes_per_sweep"] for r in recs],
            "per_rank_achieved_GBps": [round(x, 1) for x in per_rank_gbps],
            "per_rank_frac_of_8TBps": [round(x / HBM_PEAK_GBS, 4) for x in per_rank_gbps],
            "aggregate_GBps": round(abytes / (max(ev) * 1e-3) / 1e9, 1),
            "frac_of_aggregate_peak": round(abytes / (max(ev) * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "per_sweep_ms_rank0": recs[0].get("per_sweep_ms"),
            "kernel": recs[0].get("kernel"), "placement_probe_ms_rank0": recs[0].get("placement_probe_ms"),
            "seconds": max(r["seconds"] for r in recs)}


def supervisor_line(a, world, pre, results, ladder, good):
    """The ONE JSON line of an N > 1 run (rank 0's supervisor)."""
    S_name = {"rccl": "native amt_slab_* (C++ runtime, ncclSend/ncclRecv)", "ipc": "native amt_slab_* (C++ runtime, IPC peer copies + mailbox)",
              "torch-rccl": "torch.distributed P2P (patch.SlabStepper)"}
    transports = {k: summarize_rung(a, world, recs) for k, recs in results.items()}
    head_kind = next((k for k in ("rccl", "ipc", "torch-rccl") if k in good), None)
    head = transports.get(head_kind) if head_kind else None
    ndev = pre[0].get("devices") or 0
    itemsize = 8 if a.dtype == "f64" else 4
    abytes = algorithmic_bytes(a.ni, a.nk, a.nj, itemsize)
    out = {
        "metric": "advance_mu_t grid-cells/sec (Mcells/s) + achieved HBM GB/s",
        "value": head["value"] if head else None, "unit": "Mcells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": head["ms_per_step"] if head else None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic (seeded closed-form WRF-shaped fields, include/amt_synth.h)",
        "config": {"workload": f"advance_mu_t {a.ni}x{a.nk}x{a.nj} (i,k,j) {a.dtype}, {world} j-slab(s), one-row halo exchange per sweep",
                   "ni": a.ni, "nk": a.nk, "nj": a.nj, "variant": a.variant, "halo_overlap": not a.no_overlap,
                   "halo_transport": head["halo_transport"] if head else None,
                   "ranks_share_a_device": bool(ndev and world > ndev),
                   "halo_pull": head.get("halo_pull") if head else None,
                   "halo_schedule": None if not head or a.cpu_dry_run else
                                    ("no overlap" if a.no_overlap else
                                     ("host-waited: post, interior on its own, pull + boundary rows behind it" if head_kind == "ipc" and os.environ.get("AMT_IPC_HOST_WAIT", "1") != "0"
                                      else "device-waited: exchange beside the interior (2 rounds unless --beside-rounds)")),
                   "interior_plan": {"beside_rounds": a.beside_rounds or 2, "beside_reserve_cus": a.beside_reserve},
                   "halo_bytes_per_rank_per_sweep": head["halo_bytes_per_rank_per_sweep"][0] if head else None,
                   "kernel": head.get("kernel") if head else None},
        "stepper": (S_name.get(head_kind) if not a.cpu_dry_run else head["stepper"] + " [dry run]") if head_kind else None,
        "value_transport": head_kind,
        "ranks_seen": head["ranks_seen"] if head else 0,
        "launched_by": "bench.py self-launch" if os.environ.get("AMT_BENCH_SELF_LAUNCHED") else "external launcher",
        "verified_vs_oracle": (None if not head or head["verified_first_sweep"] is None else
                               bool(head["verified_first_sweep"] and head["verified_later_sweep_after_new_inputs"])),
        "transports": transports,
        "ladder": ladder,
        "preflight": {"devices": ndev, "device_name": pre[0].get("device_name"), "peer_access": pre[0].get("peer_access"),
                      "rccl": pre[0].get("rccl"), "dev_shm_free_MiB": pre[0].get("dev_shm_free_MiB"),
                      "HSA_ENABLE_IPC_MODE_LEGACY": pre[0].get("HSA_ENABLE_IPC_MODE_LEGACY")},
    }
    if head:
        ev_max = head["rank_event_ms_per_step_min_max"][1]
        achieved = abytes / world / (ev_max * 1e-3) / 1e9                       # GB/s per GPU, slowest rank
        out["rank_ms_per_step_min_max"] = head["rank_event_ms_per_step_min_max"]
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                           "algorithmic_bytes_per_launch": abytes // world, "kernel_ms_per_launch": round(ev_max, 4),
                           "per_rank_achieved_GBps": head["per_rank_achieved_GBps"], "per_rank_frac": head["per_rank_frac_of_8TBps"],
                           "aggregate_GBps": head["aggregate_GBps"], "aggregate_peak_GBps": HBM_PEAK_GBS * world,
                           "frac_of_aggregate_peak": head["frac_of_aggregate_peak"],
                           "note": "per rank: the algorithmic bytes of the rank's own rows / its HIP-event time per sweep (interior + exchange "
                                   "+ boundary rows), against 8 TB/s; `achieved` / `frac` are the slowest rank's share of an even split"}
    if a.cpu_dry_run:
        gb_rows = [r["rows"][1] - r["rows"][0] + 1 for r in results[head_kind]] if head_kind else None
        out.update(dry_run=True, note="CPU plumbing run of the N > 1 path with an injected compute callable: not a measurement")
        out["config"]["rows_per_rank"] = gb_rows
    elif ndev and world > ndev:
        out["note"] = (f"{world} ranks share {ndev} device(s): a correctness run of the whole N > 1 path (slabs, halo exchange, "
                       "verification, reductions); `value` is NOT a scaling measurement")
    return out


def main():
    a = parse()
    if a.rung_child:
        return rung_child(a)
    if a.emulate_world > 1:
        return emulate_one_rank(a)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and a.stepper == "native" and a.backend == "nccl":
        return supervise(a)             # the ladder of transports, every rung in fresh child processes
    if a.cpu_dry_run:
        raise SystemExit("bench.py --cpu-dry-run serves the N > 1 ladder only (--gpus N, native stepper)")
    return run_rank(a)                  # N = 1, and the in-process bring-up modes (--backend gloo, --stepper torch)


if __name__ == "__main__":
    main()
