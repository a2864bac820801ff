#!/usr/bin/env python3
"""Per-instantiation resources of the march kernels, from the compiler's own remarks
(`make -C csrc resources` -> csrc/build/march_resources.txt, hipcc -Rpass-analysis=kernel-resource-usage):
VGPRs, scratch (spills) per lane, SGPRs, occupancy.  rocprofv3's kernel-trace CSV reports neither the
real VGPR allocation nor dynamic LDS for these kernels, so tracked evidence takes them from here.

  kernel_resources.py [--json] [--check]     --check: exit 1 if an instantiation the launcher can
                                             select has scratch (see SELECTABLE below)
"""
from __future__ import annotations

import json
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "csrc"
REMARKS = CSRC / "build" / "march_resources.txt"


def parse(path: Path = REMARKS):
    txt = path.read_text()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    rows = []
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else None
        m = re.match(r"(?:void )?(\w+)<(.*)>\(", d)
        rows.append({"kernel": m.group(1) if m else d, "targs": m.group(2) if m else "",
                     "vgprs": g("VGPRs"), "agprs": g("AGPRs"), "sgprs": g("SGPRs"),
                     "scratch_bytes_per_lane": g(r"ScratchSize \[bytes/lane\]"),
                     "occupancy_waves_per_simd": g(r"Occupancy \[waves/SIMD\]"),
                     "static_lds_bytes": g(r"LDS Size \[bytes/block\]")})
    return rows


def main():
    if not REMARKS.exists():
        subprocess.run(["make", "-C", str(CSRC), "resources"], check=True, capture_output=True)
    rows = parse()
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
    else:
        for r in rows:
            print(f"{r['kernel']:24s} <{r['targs']:28s}> vgpr {r['vgprs']:4d} scratch {r['scratch_bytes_per_lane']:4d} "
                  f"sgpr {r['sgprs']:3d} occ {r['occupancy_waves_per_simd']}")
    if "--check" in sys.argv:
        bad = [r for r in rows if r["scratch_bytes_per_lane"] and selectable(r)]
        for r in bad:
            print(f"SPILLS: {r['kernel']}<{r['targs']}> scratch {r['scratch_bytes_per_lane']} B/lane", file=sys.stderr)
        # a selectable name that matches no compiled instantiation would make the check pass vacuously
        compiled = {f"{r['kernel']}<{r['targs']}>" for r in rows}
        missing = sorted(selectable_names() - compiled)
        for name in missing:
            print(f"NOT IN THE COMPILER'S REMARKS (name mismatch or stale build/march_resources.txt): {name}", file=sys.stderr)
        sys.exit(1 if bad or missing else 0)


def selectable_names():
    """The instantiations amt_launch_march can pick without an AMT_MARCH_* override, from the library
    itself (amt_march_selectable: the launcher's own preference lists run over level counts, layouts and
    launch sizes -- pure host logic, no GPU needed)."""
    import ctypes
    lib = CSRC.parent / "libamt_advance_mu_t.so"
    L = ctypes.CDLL(str(lib))
    L.amt_march_selectable.restype = ctypes.c_int
    L.amt_march_selectable.argtypes = [ctypes.c_char_p, ctypes.c_int]
    n = L.amt_march_selectable(None, 0)
    buf = ctypes.create_string_buffer(n)
    L.amt_march_selectable(buf, n)
    return set(buf.value.decode().split("\n")) - {""}


_names = None


def selectable(r) -> bool:
    global _names
    if _names is None:
        _names = selectable_names()
    return f"{r['kernel']}<{r['targs']}>" in _names


if __name__ == "__main__":
    main()
