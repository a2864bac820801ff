"""Is a slow bench line a slow BOX or a slow PLACEMENT?  One process, one GPU (run through gpurun):

  1. clocks / power state as rocm-smi reports them (before torch touches the GPU);
  2. a streaming-copy ceiling (amt_calib_stream_copy, 16 B per lane, 2 x 4 GiB) -- the box's own
     HBM rate in this process;
  3. the bench workload on the arrays as torch's allocator places them (K fresh placements:
     the allocator's cache is emptied and a spacer of another size is put in front each time);
  4. the same workload on ten 3-D arrays carved out of ONE arena at chosen staggers between
     consecutive arrays (relative placement of the eleven streams, bits below 2 MiB);
  5. row-length variants (idim + extra) on separate allocations.

usage: python profiles/box_probe.py [--placements 4] [--staggers 0 4096 ...] [--extras 0 64 160 224]
Every line printed is JSON; the last one is the summary.
"""
import argparse
import ctypes
import json
import subprocess
import sys
from pathlib import Path


def smi():
    out = {}
    for flag in ("--showclocks", "--showpower", "--showperflevel", "--showmaxpower", "--showtemp"):
        try:
            r = subprocess.run(["rocm-smi", flag, "--json"], capture_output=True, text=True, timeout=30)
            out[flag] = json.loads(r.stdout) if r.stdout.strip().startswith("{") else r.stdout[-400:]
        except Exception as e:  # noqa: BLE001
            out[flag] = f"{type(e).__name__}: {e}"
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--ni", type=int, default=4096)
ap.add_argument("--nk", type=int, default=60)
ap.add_argument("--nj", type=int, default=4096)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--placements", type=int, default=4)
ap.add_argument("--staggers", type=int, nargs="*", default=[0, 256, 1024, 4096, 16384, 65536, 262144, 1048576 + 4096])
ap.add_argument("--extras", type=int, nargs="*", default=[0, 64, 160, 224])
a = ap.parse_args()
smi_before = smi()
print(json.dumps({"smi_before": smi_before}), flush=True)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
L = pkg.load_library()
S = pkg.synth
dtype = np.float64 if a.dtype == "f64" else np.float32
W = np.dtype(dtype).itemsize
abytes = W * a.ni * a.nj * (11 * a.nk + 14)


def timed(call, reps=4, rounds=3):
    call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts), max(ts)


def copy_ceiling(nbytes=4 << 30):
    src = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda").fill_(1.0)
    dst = torch.empty_like(src)
    s = torch.cuda.current_stream().cuda_stream

    def call():
        rc = L.amt_calib_stream_copy(ctypes.c_void_p(s), ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()),
                                     ctypes.c_size_t(nbytes), 16)
        assert rc == 0
    lo, hi = timed(call, reps=4, rounds=3)
    del src, dst
    return 2 * nbytes / lo / 1e6, 2 * nbytes / hi / 1e6      # GB/s (read + write)


summary = {}
c_hi, c_lo = copy_ceiling()
summary["copy_GBps_best_worst"] = [round(c_hi, 1), round(c_lo, 1)]
print(json.dumps({"copy_ceiling_GBps": summary["copy_GBps_best_worst"]}), flush=True)

# 3. placements as the allocator gives them
b = S.domain_bounds(a.ni, a.nk, a.nj, aligned=True)
place = []
spacer = None
for k in range(a.placements):
    torch.cuda.empty_cache()
    if k:
        spacer = torch.empty((k * 1237 + 311) << 20, dtype=torch.uint8, device="cuda")    # shifts what follows
    dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
    lo, hi = timed(pkg.bind_device_call(*dev.args()))
    bases = {n: dev.arrays[n].data_ptr() for n in S.RANK3}
    # does a plain read-only stream over the SAME pages see the placement too?  (then it is the memory side,
    # not this kernel's access pattern)
    sink = torch.zeros(8, dtype=torch.float64, device="cuda")
    stream_rates = {}
    for n in ("u", "t_1", "ft", "ww_1"):
        t = dev.arrays[n]
        nb = t.numel() * t.element_size() // 16 * 16
        s_ = torch.cuda.current_stream().cuda_stream

        def rd(t=t, nb=nb):
            assert L.amt_calib_stream_rate(ctypes.c_void_p(s_), ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(t.data_ptr()),
                                           ctypes.c_size_t(nb), 1) == 0
        rlo, _ = timed(rd, reps=4, rounds=2)
        stream_rates[n] = round(nb / rlo / 1e6, 1)
    place.append({"placement": k, "ms_min": round(lo, 3), "ms_max": round(hi, 3), "frac": round(abytes / lo / 1e6 / 8000, 4),
                  "read_stream_GBps": stream_rates,
                  "base_mod_2MiB": {n: p % (1 << 21) for n, p in bases.items()},
                  "base_GiB": {n: round(p / 2**30, 3) for n, p in bases.items()}})
    print(json.dumps(place[-1]), flush=True)
    if k < a.placements - 1:
        del dev
    spacer = None
summary["placements_ms"] = [p["ms_min"] for p in place]

# 4. arena staggers (the last placement's data is the source)
base = dev
order = list(S.RANK3) + [n for n in S.FIELD_NAMES if n not in S.RANK3]
sizes = {n: base.arrays[n].numel() * base.arrays[n].element_size() for n in order}
if a.staggers:
    smax = max(a.staggers)
    M2 = 1 << 21
    total = sum(-(-sz // M2) * M2 + smax for sz in sizes.values()) + 2 * M2
    arena = torch.empty(total, dtype=torch.uint8, device="cuda:0")
    stag = []
    for stagger in a.staggers:
        off = (-arena.data_ptr()) % M2
        arrays = {}
        for n in order:
            src = base.arrays[n]
            view = arena[off:off + sizes[n]].view(src.dtype).view(src.shape)
            view.copy_(src)
            arrays[n] = view
            off += -(-sizes[n] // M2) * M2 + (stagger if n in S.RANK3 else 0)
        devA = S.Patch(base.bounds, base.config, arrays, base.rdx, base.rdy, base.dts, base.epssm, base.global_dims)
        lo, hi = timed(pkg.bind_device_call(*devA.args()))
        stag.append({"stagger": stagger, "ms_min": round(lo, 3), "ms_max": round(hi, 3)})
        print(json.dumps(stag[-1]), flush=True)
    summary["arena_stagger_ms"] = {str(s["stagger"]): s["ms_min"] for s in stag}
    del arena, arrays, devA
del base, dev
torch.cuda.empty_cache()

# 5. row lengths
rows = []
for extra in a.extras:
    bb = b.replace(ime=b.ime + extra)
    dev = S.make_patch(bb, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
    lo, hi = timed(pkg.bind_device_call(*dev.args()))
    rows.append({"idim": bb.idim, "ms_min": round(lo, 3), "ms_max": round(hi, 3)})
    print(json.dumps(rows[-1]), flush=True)
    del dev
    torch.cuda.empty_cache()
summary["idim_ms"] = {str(r["idim"]): r["ms_min"] for r in rows}
c_hi, c_lo = copy_ceiling()
summary["copy_GBps_after"] = [round(c_hi, 1), round(c_lo, 1)]
summary["smi_after"] = smi()
print(json.dumps(summary), flush=True)
