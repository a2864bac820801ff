"""Beyond the march kernel's level limit (240 fp64 / 264 fp32) AUTO runs the column kernel: what its two flavours cost there and at
WRF's level counts (AMT_COLUMN_RECOMPUTE=0: dvdxi column in LDS, one wave per 150 KB; 1: evaluated twice, nothing in LDS).
python profiles/r06_tall_columns.py"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
S = pkg.synth


def timed(call, n=5):
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


CASES = ((np.float64, 4096, 300, 256), (np.float64, 4096, 250, 256), (np.float64, 4096, 240, 256), (np.float64, 4096, 200, 256), (np.float64, 4096, 160, 512),
         (np.float64, 4096, 130, 512), (np.float64, 4096, 60, 1024), (np.float64, 4096, 20, 2048), (np.float64, 4096, 8, 2048),
         (np.float32, 4096, 320, 512), (np.float32, 4096, 300, 512), (np.float32, 4096, 280, 512), (np.float32, 4096, 264, 512), (np.float32, 4096, 250, 512),
         (np.float32, 4096, 200, 512), (np.float32, 4096, 80, 1024), (np.float32, 4096, 16, 2048))
for dtype, ni, nk, nj in CASES:
    b = S.domain_bounds(ni, nk, nj, aligned=True)
    dev = S.make_patch(b, pkg.GridConfig(), dtype=dtype, seed=1, device="cuda:0")
    w = 8 if dtype == np.float64 else 4
    abytes = w * ni * nj * (11 * nk + 14)
    line = f"{ni}x{nk}x{nj} {np.dtype(dtype).name}:"
    for force in ("0", "1"):
        os.environ["AMT_COLUMN_RECOMPUTE"] = force
        ms = timed(pkg.bind_device_call(*dev.args(), variant=pkg.VARIANT_COLUMN))
        line += f"  column kernel {'recompute' if force == '1' else 'LDS column'} {ms:8.3f} ms = {abytes / ms / 1e6 / 8000:.3f} of 8 TB/s;"
    os.environ.pop("AMT_COLUMN_RECOMPUTE")
    ms = timed(pkg.bind_device_call(*dev.args()))
    line += f"  AUTO {ms:8.3f} ms = {abytes / ms / 1e6 / 8000:.3f} ({pkg.load_library().amt_march_last_kernel().decode()[:52]})"
    print(line, flush=True)
    del dev
    torch.cuda.empty_cache()
