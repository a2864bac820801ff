"""Known-byte-count streaming copies (4, 8, 16 B per lane) for calibrating FETCH_SIZE / WRITE_SIZE."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
L = pkg.load_library()
NBYTES = 2 << 30        # 2 GiB read + 2 GiB written per launch, far beyond the 256 MiB Infinity Cache
src = torch.empty(NBYTES // 8, dtype=torch.float64, device="cuda").normal_()
dst = torch.empty_like(src)
torch.cuda.synchronize()
s = torch.cuda.current_stream().cuda_stream
for width in (4, 8, 16, 4, 8, 16):
    rc = L.amt_calib_stream_copy(ctypes.c_void_p(s), ctypes.c_void_p(dst.data_ptr()),
                                 ctypes.c_void_p(src.data_ptr()), NBYTES, width)
    assert rc == 0
torch.cuda.synchronize()
print("calib ok", NBYTES)
