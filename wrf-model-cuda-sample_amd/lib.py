"""ctypes binding of the C-ABI declared in include/amt_advance_mu_t.h.

This is the "reference-side binding" a Python host would add (INTEGRATION.md shows the
Fortran ISO_C_BINDING and C ones).  Loading fails loudly when the HIP library has not
been built -- there is no fallback implementation.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
_LIB_NAME = "libamt_advance_mu_t.so"

# status codes of include/amt_advance_mu_t.h
OK, ERR_HIP, ERR_PRECONDITION, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_ALLOC, ERR_COMM = range(7)


class AmtError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"amt status {status}: {message}")
        self.status = status


def library_path() -> Path:
    """The in-tree HIP library; AMT_LIBRARY overrides it (A/B builds of the kernels)."""
    import os
    override = os.environ.get("AMT_LIBRARY")
    return Path(override) if override else PKG_DIR / _LIB_NAME


_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long


def _adv_sig(real, device: bool):
    head = [_P, _I] if device else []
    return head + [_P] * 18 + [real] * 4 + [_P] * 8 + [_I] * (3 + 17)


# every symbol include/amt_advance_mu_t.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "amt_version": (ctypes.c_char_p, []),
    "amt_status_string": (ctypes.c_char_p, [_I]),
    "amt_last_error": (ctypes.c_char_p, []),
    "amt_device_count": (_I, []),
    "amt_advance_mu_t_f32": (_I, _adv_sig(ctypes.c_float, False)),
    "amt_advance_mu_t_f64": (_I, _adv_sig(ctypes.c_double, False)),
    "amt_advance_mu_t_device_f32": (_I, _adv_sig(ctypes.c_float, True)),
    "amt_advance_mu_t_device_f64": (_I, _adv_sig(ctypes.c_double, True)),
    "amt_compute_window": (_I, [_I] * 13 + [ctypes.POINTER(_I)] * 6),
    "amt_domain_create": (_I, [ctypes.POINTER(_P), _I] + [_I] * 20),
    "amt_domain_wrap": (_I, [ctypes.POINTER(_P), _I] + [_I] * 20 + [ctypes.POINTER(_P), _P]),
    "amt_domain_destroy": (_I, [_P]),
    "amt_domain_set_scalars": (_I, [_P] + [ctypes.c_double] * 4),
    "amt_domain_set_variant": (_I, [_P, _I]),
    "amt_domain_upload": (_I, [_P, _I, _P]),
    "amt_domain_download": (_I, [_P, _I, _P]),
    "amt_domain_upload_rows": (_I, [_P, _I, _I, _I, _P]),
    "amt_domain_download_rows": (_I, [_P, _I, _I, _I, _P]),
    "amt_domain_fill_synthetic": (_I, [_P, ctypes.c_uint64] + [_L] * 6),
    "amt_domain_fill_fields": (_I, [_P, ctypes.c_uint64, ctypes.c_uint64] + [_L] * 6),
    "amt_domain_poison_halos": (_I, [_P, _I]),
    "amt_domain_step": (_I, [_P, _I]),
    "amt_domain_tune_placement": (_I, [_P, _I, ctypes.POINTER(ctypes.c_float)]),
    "amt_domain_placement": (_I, [_P, ctypes.POINTER(ctypes.c_float), _I]),
    "amt_domain_step_timed": (_I, [_P, _I, ctypes.POINTER(ctypes.c_float)]),
    "amt_domain_sync": (_I, [_P]),
    "amt_domain_field_ptr": (_P, [_P, _I]),
    "amt_domain_stream": (_P, [_P]),
    "amt_synth_fill_host": (_I, [_I, _I, _P, ctypes.c_uint64] + [_L] * 9),
    "amt_synth_fill_device": (_I, [_P, _I, _I, _P, ctypes.c_uint64] + [_L] * 9),
    "amt_calib_stream_copy": (_I, [_P, _P, _P, ctypes.c_size_t, _I]),
    "amt_calib_stream_rate": (_I, [_P, _P, _P, ctypes.c_size_t, _I]),
    "amt_host_pin": (_I, [_P, ctypes.c_size_t]),
    "amt_host_unpin": (_I, [_P]),
    "amt_host_release": (_I, []),
    "amt_host_set_devices": (_I, [_I, ctypes.POINTER(_I)]),
    "amt_host_devices": (_I, [ctypes.POINTER(_I), _I]),
    "amt_host_cache_enable": (_I, [_I]),
    "amt_host_cache_check": (_I, [_I]),
    "amt_host_invalidate": (_I, [_P]),
    "amt_host_defer": (_I, [_P, _I]),
    "amt_host_fetch": (_I, [_P]),
    "amt_host_stale": (_I, [_P]),
    "amt_set_device": (_I, [_I]),
    "amt_comm_unique_id": (_I, [_P]),
    "amt_comm_rendezvous_file": (_I, [ctypes.c_char_p, ctypes.c_uint64, _I, _I, ctypes.c_double, _P]),
    "amt_comm_launch_nonce": (ctypes.c_uint64, []),
    "amt_slab_create": (_I, [ctypes.POINTER(_P), _P, _I, _I, _P, _I]),
    "amt_slab_destroy": (_I, [_P]),
    "amt_slab_exchange": (_I, [_P]),
    "amt_slab_step": (_I, [_P, _I]),
    "amt_slab_step_timed": (_I, [_P, _I, ctypes.POINTER(ctypes.c_float)]),
    "amt_slab_sync": (_I, [_P]),
    "amt_slab_transport": (ctypes.c_char_p, [_P]),
    "amt_slab_pull_mode": (ctypes.c_char_p, [_P]),
    "amt_slab_set_skew_us": (_I, [_P, _I]),
    "amt_slab_halo_bytes": (_L, [_P]),
    "amt_slab_comm_info": (_I, [_P, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "amt_slab_barrier": (_I, [_P]),
    "amt_slab_max": (_I, [_P, ctypes.POINTER(ctypes.c_double)]),
    "amt_grid_create": (_I, [ctypes.POINTER(_P), _P, _I, _I, _I, _I, _P, _I]),
    "amt_grid_destroy": (_I, [_P]),
    "amt_grid_exchange": (_I, [_P]),
    "amt_grid_step": (_I, [_P, _I]),
    "amt_grid_step_timed": (_I, [_P, _I, ctypes.POINTER(ctypes.c_float)]),
    "amt_grid_sync": (_I, [_P]),
    "amt_grid_set_skew_us": (_I, [_P, _I]),
    "amt_grid_halo_bytes": (_L, [_P]),
    "amt_grid_transport": (ctypes.c_char_p, [_P]),
    "amt_grid_pull_mode": (ctypes.c_char_p, [_P]),
    "amt_grid_comm_info": (_I, [_P, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "amt_grid_barrier": (_I, [_P]),
    "amt_grid_max": (_I, [_P, ctypes.POINTER(ctypes.c_double)]),
    "amt_march_force_shape": (_I, [_I] * 7),
    "amt_march_rows_for": (_I, [ctypes.c_long, _I, _I, ctypes.c_long, _I, _I]),
    "amt_march_set_xchunk": (_I, [_I]),
    "amt_march_set_beside": (_I, [_I, _I]),
    "amt_march_set_stream_policy": (_I, [_I]),
    "amt_march_last_kernel": (ctypes.c_char_p, []),
    "amt_march_selectable": (_I, [ctypes.c_char_p, _I]),
}


def load_library() -> ctypes.CDLL:
    """Load the HIP library (once).  torch, when used in the same process, must be imported
    first so that both share one HIP runtime (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not path.exists():
            raise AmtError(ERR_NO_DEVICE, f"{path} not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                                          "(make -C wrf-model-cuda-sample_amd/csrc); there is no CPU fallback")
        try:
            # one HIP runtime per process: torch ships its own libamdhip64.so.7 and must bring it in
            # before this library resolves the same SONAME from /opt/rocm (otherwise torch ends up on
            # a runtime its other bundled libraries do not match: "no ROCm-capable device")
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(str(path))
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != OK:
        L = load_library()
        msg = L.amt_last_error().decode() or L.amt_status_string(status).decode()
        raise AmtError(status, msg)
