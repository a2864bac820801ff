"""The reference drivers' on-disk format: one raw stream file per variable, every element a
big-endian 4-byte int / IEEE float, arrays i-fastest over the full memory extent
(advance_mu_t_driver.f90:330,364; common.cu:166-327; file names advance_mu_t_driver.c:60-219).

The reference's data set (/data2/WRFV3_Input_Output/V3.4.1/dyn_em/advance_mu_t/) is not shipped;
this module writes and reads directories in that format (fp32 only, as the format is) so that a
real dump can be replayed with ``tools/advance_mu_t_replay`` and so that the tests can fabricate
one from the synthetic inputs.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .config import GridConfig
from .synth import Bounds, Patch, INT_NAMES

# argument name -> input file name (advance_mu_t_driver.c:149-219)
INPUT_FILES = {
    "dnw": "grid_dnw.bin", "fnm": "grid_fnm.bin", "fnp": "grid_fnp.bin", "rdnw": "grid_rdnw.bin",
    "mut": "grid_mut.bin", "muu": "grid_muu.bin", "muv": "grid_muv.bin", "mu_tend": "mu_tend.bin",
    "msfuy": "grid_msfuy.bin", "msfvx_inv": "grid_msfvx_inv.bin", "msfty": "grid_msfty.bin",
    "msftx": "grid_msftx.bin", "mu": "grid_mu_2.bin",
    "u": "grid_u_2.bin", "u_1": "grid_u_save.bin", "v": "grid_v_2.bin", "v_1": "grid_v_save.bin",
    "t_1": "grid_t_save.bin", "ft": "t_tend.bin",
    "ww": "grid_ww.bin", "ww_1": "ww1.bin", "t": "grid_t_2.bin", "t_ave": "t_2save.bin",
}
# argument name -> golden output file name (advance_mu_t_driver.f90:224-231)
OUTPUT_FILES = {
    "ww": "grid_ww_output.bin", "ww_1": "ww1_output.bin", "t": "grid_t_2_output.bin",
    "t_ave": "t_2save_output.bin", "mu": "grid_mu_2_output.bin", "muave": "muave_output.bin",
    "muts": "grid_muts_output.bin", "mudf": "grid_mudf_output.bin",
}
SCALAR_FILES = {"rdx": "grid_rdx.bin", "rdy": "grid_rdy.bin", "dts": "dts_rk.bin", "epssm": "grid_epssm.bin"}
FLAG_FILES = {"nested": "config_flags_nested.bin", "periodic_x": "config_flags_periodic_x.bin",
              "specified": "config_flags_specified.bin"}


def _w(path: Path, a, dtype):
    np.ascontiguousarray(a).astype(dtype).tofile(path)


def write_inputs(directory, patch: Patch, with_kds: bool = True) -> None:
    """Write the input side of a dump directory from an fp32 host Patch."""
    d = Path(directory)
    d.mkdir(parents=True, exist_ok=True)
    b = patch.bounds
    for n in INT_NAMES:
        _w(d / f"{n}.bin", np.array([getattr(b, n)]), ">i4")
    if with_kds:                                   # the C / CUDA drivers also read kds (advance_mu_t_driver.c:64)
        _w(d / "kds.bin", np.array([b.kts]), ">i4")
    for n, f in SCALAR_FILES.items():
        _w(d / f, np.array([getattr(patch, n)], dtype=np.float32), ">f4")
    for n, f in FLAG_FILES.items():
        _w(d / f, np.array([int(getattr(patch.config, n))]), ">i4")
    for n, f in INPUT_FILES.items():
        a = patch.arrays[n]
        if a.dtype != np.float32:
            raise TypeError("the dump format holds 4-byte reals")
        _w(d / f, a, ">f4")


def write_outputs(directory, patch: Patch) -> None:
    """Write the eight golden '*_output.bin' files from an (updated) fp32 host Patch."""
    d = Path(directory)
    d.mkdir(parents=True, exist_ok=True)
    for n, f in OUTPUT_FILES.items():
        _w(d / f, patch.arrays[n], ">f4")


def read_inputs(directory) -> Patch:
    """Read a dump directory back into a Patch (OUT arrays muave, muts, mudf zero-filled)."""
    d = Path(directory)
    ints = {n: int(np.fromfile(d / f"{n}.bin", dtype=">i4")[0]) for n in INT_NAMES}
    b = Bounds(**ints)
    cfg = GridConfig(**{n: bool(np.fromfile(d / f, dtype=">i4")[0]) for n, f in FLAG_FILES.items()})
    arrays = {}
    for n, f in INPUT_FILES.items():
        arrays[n] = np.fromfile(d / f, dtype=">f4").astype(np.float32).reshape(b.shape(n))
    for n in ("muave", "muts", "mudf"):
        arrays[n] = np.zeros(b.shape(n), dtype=np.float32)
    sc = {n: float(np.fromfile(d / f, dtype=">f4")[0]) for n, f in SCALAR_FILES.items()}
    return Patch(b, cfg, arrays, sc["rdx"], sc["rdy"], sc["dts"], sc["epssm"])


def read_outputs(directory, bounds: Bounds) -> dict:
    d = Path(directory)
    return {n: np.fromfile(d / f, dtype=">f4").astype(np.float32).reshape(bounds.shape(n))
            for n, f in OUTPUT_FILES.items()}
