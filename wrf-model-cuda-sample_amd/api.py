"""advance_mu_t -- Python mirror of the reference routine's interface.

Same name, argument order and meaning as ``SUBROUTINE advance_mu_t``
(module_small_step_em.f90:7-18); ``config_flags`` is anything with
``.periodic_x/.specified/.nested`` (``GridConfig``).  Arrays are updated in place.

* numpy arrays  -> one-shot host drop-in  ``amt_advance_mu_t_f32/_f64``
* torch CUDA tensors -> device-resident drop-in ``amt_advance_mu_t_device_f32/_f64`` on
  ``stream`` (default: torch's current stream), asynchronous.

Arrays are i-fastest: a 3-D field is a C-contiguous array of shape (jdim, kdim, idim)
(= Fortran (ims:ime, kms:kme, jms:jme)), a 2-D field (jdim, idim), a 1-D field (kdim,).
"""
from __future__ import annotations

import ctypes
import threading

import numpy as np

from . import lib as _lib
from .config import flags_as_ints

VARIANT_AUTO, VARIANT_COLUMN, VARIANT_MARCH = 0, 1, 2
LAUNCH_BESIDE_OTHERS = 0x100      # OR into `variant`: another stream's kernels run beside this launch (include/amt_advance_mu_t.h)


def compute_window(config_flags, ids, ide, jds, jde, its, ite, jts, jte, kts, kte):
    """(i_start, i_end, j_start, j_end, k_start, k_end) of module_small_step_em.f90:91-106."""
    L = _lib.load_library()
    out = [ctypes.c_int() for _ in range(6)]
    _lib.check(L.amt_compute_window(*flags_as_ints(config_flags), ids, ide, jds, jde,
                                    its, ite, jts, jte, kts, kte, *[ctypes.byref(o) for o in out]))
    return tuple(o.value for o in out)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _expected_sizes(ims, ime, jms, jme, kms, kme):
    """Element counts of the 26 array arguments, in argument order (18 before the scalars, 8 after),
    for the memory extents ims:ime, kms:kme, jms:jme (module_small_step_em.f90:30-64)."""
    idim, kdim, jdim = ime - ims + 1, kme - kms + 1, jme - jms + 1
    rank3 = {0, 1, 2, 3, 4, 5, 13, 14, 15, 16}          # ww ww_1 u u_1 v v_1 | t t_1 t_ave ft
    want = [(jdim * kdim * idim) if n in rank3 else (jdim * idim) for n in range(18)]
    return want + [kdim] * 4 + [jdim * idim] * 4


def bind_device_call(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                     t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,
                     msfuy, msfvx_inv, msftx, msfty, config_flags,
                     ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,
                     its, ite, jts, jte, kts, kte, *, stream=None, variant=VARIANT_AUTO):
    """Validate once and return a zero-argument callable that enqueues this device-resident
    advance_mu_t call (same tensors, same bounds) -- for per-sub-step loops, where rebuilding the
    48 ctypes arguments every sweep would make the host the bottleneck."""
    import torch
    L = _lib.load_library()
    arrays = (ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1, t_ave, ft, mu_tend,
              dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty)
    dt = ww.dtype
    if dt not in (torch.float32, torch.float64):
        raise TypeError(f"unsupported dtype {dt}")
    # the kernel gets raw pointers: a tensor of another extent (a slab patch bound with the whole
    # domain's bounds, say) would be read or written out of bounds on the device
    for a, n in zip(arrays, _expected_sizes(ims, ime, jms, jme, kms, kme)):
        if not (_is_torch(a) and a.is_cuda and a.dtype == dt and a.is_contiguous() and a.numel() == n):
            raise TypeError("device call needs contiguous CUDA tensors of one dtype and of the memory extents")
    real = ctypes.c_float if dt == torch.float32 else ctypes.c_double
    fn = L.amt_advance_mu_t_device_f32 if dt == torch.float32 else L.amt_advance_mu_t_device_f64
    if stream is None:
        stream = torch.cuda.current_stream(ww.device)
    handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    cargs = ([ctypes.c_void_p(handle), int(variant)]
             + [ctypes.c_void_p(a.data_ptr()) for a in arrays[:18]]
             + [real(float(x)) for x in (rdx, rdy, dts, epssm)]
             + [ctypes.c_void_p(a.data_ptr()) for a in arrays[18:]]
             + list(flags_as_ints(config_flags))
             + [int(x) for x in (ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte)])
    device_index = ww.device.index if ww.device.index is not None else torch.cuda.current_device()
    keep = arrays                                   # keeps the tensors alive as long as the closure

    def call():
        if torch.cuda.current_device() != device_index:
            torch.cuda.set_device(device_index)
        status = fn(*cargs)
        if status:
            _lib.check(status)
        return keep is None
    return call


def advance_mu_t(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
                 t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,
                 msfuy, msfvx_inv, msftx, msfty, config_flags,
                 ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,
                 its, ite, jts, jte, kts, kte, *, stream=None, variant=VARIANT_AUTO):
    L = _lib.load_library()
    arrays_a = (ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1, t_ave, ft, mu_tend)
    arrays_b = (dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty)
    ints = [int(x) for x in (ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte)]
    want = _expected_sizes(ims, ime, jms, jme, kms, kme)
    flags = list(flags_as_ints(config_flags))

    if _is_torch(ww):
        import torch
        dt = ww.dtype
        if dt not in (torch.float32, torch.float64):
            raise TypeError(f"unsupported dtype {dt}")
        for a, n in zip(arrays_a + arrays_b, want):
            if not (_is_torch(a) and a.is_cuda and a.dtype == dt and a.is_contiguous() and a.numel() == n):
                raise TypeError("device call needs contiguous CUDA tensors of one dtype and of the memory extents")
        real = ctypes.c_float if dt == torch.float32 else ctypes.c_double
        if stream is None:
            stream = torch.cuda.current_stream(ww.device)
        handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        fn = L.amt_advance_mu_t_device_f32 if dt == torch.float32 else L.amt_advance_mu_t_device_f64
        with torch.cuda.device(ww.device):
            status = fn(ctypes.c_void_p(handle), int(variant),
                        *[ctypes.c_void_p(a.data_ptr()) for a in arrays_a],
                        *[real(float(s)) for s in (rdx, rdy, dts, epssm)],
                        *[ctypes.c_void_p(a.data_ptr()) for a in arrays_b], *flags, *ints)
        _lib.check(status)
        return

    dt = ww.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError(f"unsupported dtype {dt}")
    for a, n in zip(arrays_a + arrays_b, want):
        if not (isinstance(a, np.ndarray) and a.dtype == dt and a.size == n
                and (a.flags["C_CONTIGUOUS"] or a.flags["F_CONTIGUOUS"])):
            raise TypeError("host call needs contiguous numpy arrays of one dtype and of the memory extents")
    real = ctypes.c_float if dt == np.float32 else ctypes.c_double
    fn = L.amt_advance_mu_t_f32 if dt == np.float32 else L.amt_advance_mu_t_f64
    try:
        status = fn(*[a.ctypes.data_as(ctypes.c_void_p) for a in arrays_a],
                    *[real(float(s)) for s in (rdx, rdy, dts, epssm)],
                    *[a.ctypes.data_as(ctypes.c_void_p) for a in arrays_b], *flags, *ints)
    finally:
        # the library may have flushed deferred outputs into the PREVIOUS call's arrays during this call (other arrays,
        # other extents): those were kept alive until now; from here on it is this call's arrays it remembers
        _remember(arrays_a + arrays_b)
    _lib.check(status)


advance_mu_t.bind = bind_device_call        # SlabStepper pre-marshals its per-sweep launches through this


# Lifetime of host arrays the library remembers.  With the residency cache or deferred outputs on, the library
# keeps the host ADDRESSES of a call's arrays after the call returns: cached inputs are recognised by address, and a deferred
# output is written through its address later (amt_host_fetch, the automatic flush on a key change / release / defer-off, the
# check mode's canaries).  A C or Fortran host owns its arrays for the run; numpy arrays are garbage-collected -- so this
# module holds strong references to the arrays of the calling thread's LAST one-shot call for as long as either mode is on in
# that thread, and drops them when both are off and nothing is stale.
_tl = threading.local()


def _remember(arrays) -> None:
    if getattr(_tl, "cache_on", False) or getattr(_tl, "defer_on", False) or _stale_any():
        _tl.held = tuple(arrays)
    else:
        _tl.held = ()


def _stale_any() -> bool:
    return int(_lib.load_library().amt_host_stale(None)) != 0        # stale, or undefined after a failed call: still remembered


def _forget_if_idle() -> None:
    if not (getattr(_tl, "cache_on", False) or getattr(_tl, "defer_on", False)) and not _stale_any():
        _tl.held = ()


def held_arrays() -> tuple:
    """The host arrays this module keeps alive for the calling thread (see above); () when none."""
    return getattr(_tl, "held", ())


def host_cache_enable(on: bool = True, check: bool = False) -> None:
    """Residency cache of the one-shot (numpy) calls of the calling thread: ww_1, u_1, v_1, t_1, ft stay on the
    device between calls (header section 1: amt_host_cache_enable).  ``check``: the checksum debug mode."""
    L = _lib.load_library()
    _lib.check(L.amt_host_cache_enable(int(bool(on))))
    _lib.check(L.amt_host_cache_check(int(bool(check))))
    _tl.cache_on = bool(on)
    _forget_if_idle()


def host_defer(array=None, on: bool = True) -> None:
    """Deferred outputs of the one-shot (numpy) calls of the calling thread (header section 1: amt_host_defer): the
    output ``array`` (None: all seven) stays on the device after a call until ``host_fetch``.

    Lifetime: the library writes the deferred values through the array's ADDRESS later (fetch, or the automatic flush when
    another set of arrays, ``host_release`` or ``host_defer(..., False)`` comes).  This module keeps the arrays of the calling
    thread's last call alive until then (``held_arrays``); arrays passed to the C-ABI directly must outlive that point."""
    L = _lib.load_library()
    ptr = None if array is None else array.ctypes.data_as(ctypes.c_void_p)
    _lib.check(L.amt_host_defer(ptr, int(bool(on))))        # defer-off fetches what is stale while the arrays are still held
    if array is None:
        _tl.defer_on = bool(on)
    elif on:
        _tl.defer_on = True
    _forget_if_idle()


def host_fetch(array=None) -> None:
    """Bring the window's cells of a deferred output (None: every stale one) down to its host array."""
    L = _lib.load_library()
    ptr = None if array is None else array.ctypes.data_as(ctypes.c_void_p)
    _lib.check(L.amt_host_fetch(ptr))
    _forget_if_idle()


def host_release() -> None:
    """Free the calling thread's device workspace (amt_host_release): stale deferred outputs come down first."""
    _lib.check(_lib.load_library().amt_host_release())
    _forget_if_idle()


def host_set_devices(device_ids=()) -> None:
    """One call, several devices (amt_host_set_devices; the reference's own host call splits j over its GPUs,
    advance_mu_t_no_async.cu:108-162): from now on every one-shot call of this thread fans its tile's rows over these device
    slots -- ids may repeat -- with halo rows from the host arrays; () turns it off."""
    ids = [int(d) for d in device_ids]
    arr = (ctypes.c_int * max(len(ids), 1))(*ids)
    _lib.check(_lib.load_library().amt_host_set_devices(len(ids), arr))


def host_devices():
    """The device slots of the calling thread ([] when its one-shot calls run on the current device only)."""
    arr = (ctypes.c_int * 64)()
    n = int(_lib.load_library().amt_host_devices(arr, 64))
    return [int(arr[k]) for k in range(n)]


def host_stale(array=None) -> bool:
    """Is the device copy of a deferred output (None: of any) newer than the host array?  Raises AmtError when the device
    copy is undefined because a call failed part-way (``host_invalidate`` makes the host array the truth again)."""
    L = _lib.load_library()
    ptr = None if array is None else array.ctypes.data_as(ctypes.c_void_p)
    v = int(L.amt_host_stale(ptr))
    if v < 0:
        raise _lib.AmtError(_lib.ERR_PRECONDITION, "the device copy of a deferred output is undefined since a call failed part-way; "
                                                   "host_invalidate(array) makes the host array the truth again")
    return v > 0


def host_invalidate(array=None) -> None:
    """The host array (numpy) changed since the last call: upload it again.  None: all cached arrays."""
    L = _lib.load_library()
    ptr = None if array is None else array.ctypes.data_as(ctypes.c_void_p)
    _lib.check(L.amt_host_invalidate(ptr))
