"""Build-time resource check of the march kernels (every selectable fp64 instantiation
beyond 60 levels had 116-416 B/lane of scratch): every instantiation the launcher can pick without an
override must be free of scratch, as reported by the compiler's own kernel-resource-usage remarks
(`make -C csrc resources`).  No GPU needed: hipcc cross-compiles, the selection logic is pure."""
import ctypes
import importlib.util
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "wrf-model-cuda-sample_amd" / "csrc"


def _tool():
    spec = importlib.util.spec_from_file_location("amt_kres", ROOT / "wrf-model-cuda-sample_amd" / "tools" / "kernel_resources.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _selectable(pkg):
    L = pkg.load_library()
    n = L.amt_march_selectable(None, 0)
    buf = ctypes.create_string_buffer(n)
    L.amt_march_selectable(buf, n)
    return [ln for ln in buf.value.decode().splitlines() if ln]


def test_selectable_instantiations_have_no_scratch(pkg):
    subprocess.run(["make", "-C", str(CSRC), "resources"], check=True, capture_output=True)
    rows = {f"{r['kernel']}<{r['targs']}>": r for r in _tool().parse()}
    names = _selectable(pkg)
    assert len(names) >= 20, names
    for name in names:
        assert name in rows, f"{name} is selectable but was not compiled"
        r = rows[name]
        assert r["scratch_bytes_per_lane"] == 0, f"{name}: {r['scratch_bytes_per_lane']} B/lane of scratch"
        wm = int(name.rstrip(">").split(", ")[-2])                 # <T, VW, KPT, HL, XD, FULL, DMA, WM, NTL>
        budget = 128 if wm == 16 else 168
        assert r["vgprs"] <= budget, (name, r["vgprs"])


def test_every_level_count_up_to_128_has_a_march_kernel(pkg):
    """NK <= 128 never falls to the column kernel in either precision on the resident layout."""
    names = "\n".join(_selectable(pkg))
    for must in ("amt_march_kernel<double, 1, 3, 1, 0, true, true, 16",      # <= 45 levels
                 "amt_march_kernel<double, 1, 4, 1, 0, true, true, 16",      # <= 60
                 "amt_march_kernel<double, 1, 3, 2, 0, true, true, 16",      # <= 90
                 "amt_march_kernel<double, 1, 4, 2, 0, false, true, 16",     # <= 120 (the general build)
                 "amt_march_kernel<double, 1, 6, 2, 0, true, true, 12",      # <= 132
                 "amt_march_kernel<float, 2, 4, 1, 0, true, true, 16",
                 "amt_march_kernel<float, 2, 3, 2, 0, true, true, 16"):
        # both cache policies of the once-read streams (last template argument: 1 = non-temporal for rows that are whole
        # 128-byte lines, 0 = plain loads for rows that are not) are selectable and compiled
        assert must + ", 1>" in names and must + ", 0>" in names, must
