"""One-off: 4096x60x8192 fp64 (164 GB resident): two rounds of 1024-row blocks in fp64 -- rows around every block seam and
both domain edges against the oracle (the helper of tests/test_gpu_00_configs.py)."""
import re
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g  # noqa: E402
import test_gpu_00_configs as T  # noqa: E402

pkg, oracle = g.load_package(), g.load_oracle()
S, L = pkg.synth, pkg.load_library()
dims = (4096, 60, 8192)
b = S.domain_bounds(*dims, aligned=True)
cfg = pkg.GridConfig(specified=True)
dev = S.make_patch(b, cfg, dtype=np.float64, seed=5150, device="cuda:0")
pkg.advance_mu_t(*dev.args()); torch.cuda.synchronize()
label = L.amt_march_last_kernel().decode()
jrows = int(re.search(r"jrows=(\d+)", label).group(1))
rows = 48
nblk = -(-(dims[2] - 2) // jrows)
starts = [1, dims[2] - rows + 1] + [2 + jrows * k - rows // 2 for k in range(1, nblk)]
checked = T._check_rows_against_oracle(pkg, oracle, dev, b, cfg, dims, np.float64, 5150, starts, rows)
print(f"{dims} f64: {label}; {nblk} blocks per tile; {len(checked)} rows in {len(starts)} chunks bit-equal to the oracle")
