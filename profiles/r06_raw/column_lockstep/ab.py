"""The column kernel's rows of a workgroup in step (AMT_COLUMN_LOCKSTEP=1: a workgroup barrier per level, so that the rows j-1, j+1 a wave
reads are the rows its siblings read at that moment) against free-running waves, both flavours, with a bit comparison of all outputs.
python profiles/r06_column_lockstep.py"""
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


OUT = ("ww", "mu", "muave", "muts", "mudf", "t", "t_ave")
CASES = ((np.float64, 4096, 300, 256), (np.float64, 4096, 250, 256), (np.float64, 4096, 130, 512), (np.float64, 4096, 60, 1024), (np.float64, 4096, 20, 2048),
         (np.float64, 4096, 8, 2048), (np.float64, 500, 60, 500), (np.float64, 64, 40, 64),
         (np.float32, 4096, 300, 512), (np.float32, 4096, 264, 512), (np.float32, 4096, 80, 1024), (np.float32, 4096, 16, 2048), (np.float32, 333, 45, 77))
for dtype, ni, nk, nj in CASES:
    b = S.domain_bounds(ni, nk, nj, aligned=(ni % 64 == 0))
    w = 8 if dtype == np.float64 else 4
    abytes = w * ni * nj * (11 * nk + 14)
    line = f"{ni}x{nk}x{nj} {np.dtype(dtype).name}:"
    for force in ("1", "0"):
        os.environ["AMT_COLUMN_RECOMPUTE"] = force
        res = {}
        for lock in ("0", "1"):
            os.environ["AMT_COLUMN_LOCKSTEP"] = lock
            dev = S.make_patch(b, pkg.GridConfig(specified=True), dtype=dtype, seed=1, device="cuda:0")
            call = pkg.bind_device_call(*dev.args(), variant=pkg.VARIANT_COLUMN)
            call()
            torch.cuda.synchronize()
            res[lock] = {n: dev.arrays[n].clone() for n in OUT}
            ms = timed(call)
            line += f"  {'recompute' if force == '1' else 'LDS'} lockstep={lock} {ms:8.3f} ms = {abytes / ms / 1e6 / 8000:.3f};"
            del dev, call
        same = all(torch.equal(res["0"][n].view(torch.int64 if w == 8 else torch.int32), res["1"][n].view(torch.int64 if w == 8 else torch.int32)) for n in OUT)
        line += f" bits {'same' if same else 'DIFFER'};"
        del res
        torch.cuda.empty_cache()
    print(line, flush=True)
