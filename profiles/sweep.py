"""One box, a FRESH process per workload (where the ten arrays land in HBM moves a big sweep by up to
10 %: in one long-lived process 4096x60x4096 fp64 measured 18.0 ms after the smaller cases had come and
gone, 15.8 in a fresh one): the launcher's own choice on a list of workloads -> a markdown table
(profiles/r02_size_sweep.md).  python profiles/sweep.py > gpurun_out/r02_size_sweep.md"""
import subprocess
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

CASES = [
    ("f64", 64, 40, 64, True), ("f64", 128, 60, 128, True), ("f64", 256, 60, 256, True), ("f64", 512, 60, 512, True),
    ("f64", 1024, 60, 1024, True), ("f64", 2048, 60, 2048, True), ("f64", 4096, 60, 512, True), ("f64", 4096, 60, 4096, True),
    ("f64", 4096, 60, 4096, False), ("f64", 4096, 20, 2048, True), ("f64", 4096, 40, 2048, True), ("f64", 4096, 45, 1024, True),
    ("f64", 4096, 76, 2048, True), ("f64", 4096, 80, 2048, True), ("f64", 4096, 88, 2048, True), ("f64", 4096, 100, 2048, True),
    ("f64", 4096, 120, 1536, True), ("f64", 4096, 128, 1536, True), ("f64", 4096, 176, 1024, True), ("f64", 4096, 264, 512, True),
    ("f32", 512, 60, 512, True), ("f32", 4096, 40, 4096, True), ("f32", 4096, 60, 4096, True), ("f32", 4095, 60, 4096, False),
    ("f32", 8192, 80, 4096, True), ("f32", 8192, 80, 8192, True), ("f32", 4096, 100, 2048, True), ("f32", 4096, 128, 2048, True),
]
if len(sys.argv) == 1:
    print("| workload | kernel the launcher picked | ms / sweep | Gcells/s | algorithmic TB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|", flush=True)
    for n in range(len(CASES)):
        r = subprocess.run([sys.executable, __file__, str(n)], capture_output=True, text=True)
        sys.stdout.write(r.stdout if r.returncode == 0 else f"| case {CASES[n]} | failed: {r.stderr[-200:]} | | | | |\n")
        sys.stdout.flush()
    sys.exit(0)

pkg = g.load_package()
L = pkg.load_library()
S = pkg.synth
for dt, ni, nk, nj, aligned in [CASES[int(sys.argv[1])]]:
    dtype = np.float64 if dt == "f64" else np.float32
    b = S.domain_bounds(ni, nk, nj, aligned=aligned)
    need = 10 * b.idim * b.kdim * b.jdim * np.dtype(dtype).itemsize * 1.05
    if torch.cuda.mem_get_info(0)[0] < need:
        print(f"| {ni}×{nk}×{nj} {dt} | skipped: needs {need / 1e9:.0f} GB | | | | |")
        continue
    dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
    call = pkg.bind_device_call(*dev.args())
    call(); call()
    torch.cuda.synchronize()
    inner = max(3, min(200, int(0.15 / max(1e-5, 1e-11 * ni * nk * nj))))
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            call()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / inner)
    ms = float(np.median(times))
    ab = np.dtype(dtype).itemsize * ni * nj * (11 * nk + 14)
    name = L.amt_march_last_kernel().decode().replace("amt_march_kernel", "")
    lay = "" if aligned else f", rows of {b.idim} elements (unaligned)"
    print(f"| {ni}×{nk}×{nj} {dt}{lay} | `{name}` | {ms:.4g} | {ni * nk * nj / ms / 1e6:.1f} | {ab / ms / 1e9:.2f} | {ab / ms / 1e9 / 8:.3f} |", flush=True)
    del dev, call
    torch.cuda.empty_cache()
