"""Seeded synthetic WRF-shaped inputs (include/amt_synth.h) and domain bookkeeping.

The reference reads a real WRF V3.4.1 dump that is not shipped
(advance_mu_t_driver.f90:38-167); here every element is a closed-form function of
(field, seed, Fortran index i,k,j), produced by ``amt_synth_fill_host`` /
``amt_synth_fill_device`` of the C-ABI -- bit-identical on host and device, and
independent of how the domain is split into patches or padded in memory.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field as _dc_field

import numpy as np

from . import lib as _lib
from .config import GridConfig

# enum amt_field (include/amt_synth.h): the Fortran argument order
FIELD_NAMES = ("ww", "ww_1", "u", "u_1", "v", "v_1", "mu", "mut", "muave", "muts", "muu", "muv",
               "mudf", "t", "t_1", "t_ave", "ft", "mu_tend", "dnw", "fnm", "fnp", "rdnw",
               "msfuy", "msfvx_inv", "msftx", "msfty")
FIELD_ID = {n: i for i, n in enumerate(FIELD_NAMES)}
RANK3 = ("ww", "ww_1", "u", "u_1", "v", "v_1", "t", "t_1", "t_ave", "ft")
RANK1 = ("dnw", "fnm", "fnp", "rdnw")
RANK2 = tuple(n for n in FIELD_NAMES if n not in RANK3 and n not in RANK1)
OUTPUTS = ("ww", "t", "t_ave", "mu", "muave", "muts", "mudf")       # written by the routine
HALO_FROM_ABOVE = ("v", "v_1", "t_1", "muv", "msfvx_inv")           # row j+1 is read (:143,:241)
HALO_FROM_BELOW = ("t_1",)                                          # row j-1 is read (:242)

RDX, RDY, DTS, EPSSM = 1.0e-3, 1.25e-3, 2.0, 0.1                    # AMT_SYNTH_* of amt_synth.h

INT_NAMES = ("ids", "ide", "jds", "jde", "kde", "ims", "ime", "jms", "jme", "kms", "kme",
             "its", "ite", "jts", "jte", "kts", "kte")


def field_rank(name: str) -> int:
    return 3 if name in RANK3 else 1 if name in RANK1 else 2


@dataclass
class Bounds:
    """The 17 integer arguments of advance_mu_t (Fortran-style inclusive)."""
    ids: int
    ide: int
    jds: int
    jde: int
    kde: int
    ims: int
    ime: int
    jms: int
    jme: int
    kms: int
    kme: int
    its: int
    ite: int
    jts: int
    jte: int
    kts: int
    kte: int

    def as_tuple(self):
        return tuple(getattr(self, n) for n in INT_NAMES)

    @property
    def idim(self):
        return self.ime - self.ims + 1

    @property
    def kdim(self):
        return self.kme - self.kms + 1

    @property
    def jdim(self):
        return self.jme - self.jms + 1

    def shape(self, name: str):
        r = field_rank(name)
        return (self.jdim, self.kdim, self.idim) if r == 3 else (self.jdim, self.idim) if r == 2 else (self.kdim,)

    def replace(self, **kw) -> "Bounds":
        d = {n: getattr(self, n) for n in INT_NAMES}
        d.update(kw)
        return Bounds(**d)


def domain_bounds(ni: int, nk: int, nj: int, *, aligned: bool = False, align_elems: int = 32) -> Bounds:
    """Bounds of an ``NI x NK x NJ`` (computed mass cells) single-patch domain, SURVEY.md
    section 8 convention: ids=jds=kds=1, ide=NI+1, jde=NJ+1, kde=NK+1, tile = domain.
    ``aligned=False``: minimal memory (0:NI+1, 1:NK+1, 0:NJ+1).
    ``aligned=True`` : i memory padded so that i = its sits ``align_elems`` elements into
    a row and rows are a multiple of ``align_elems`` long (the resident device layout).  32
    elements (256 B in fp64) measured faster than 64 (whole i-tiles, but a row stride of 4224
    elements): 16.0 vs 16.5 ms per 4096x60x4096 sweep, interleaved in one process."""
    if aligned:
        ims = 1 - align_elems
        idim = -(-(ni + 1 - ims + 1) // align_elems) * align_elems
        ime = ims + idim - 1
    else:
        ims, ime = 0, ni + 1
    return Bounds(ids=1, ide=ni + 1, jds=1, jde=nj + 1, kde=nk + 1,
                  ims=ims, ime=ime, jms=0, jme=nj + 1, kms=1, kme=nk + 1,
                  its=1, ite=ni + 1, jts=1, jte=nj + 1, kts=1, kte=nk + 1)


def slab_bounds(g: Bounds, rank: int, world: int) -> Bounds:
    """j-slab ``rank`` of ``world`` of the domain ``g``: contiguous rows jlo..jhi of the
    computed range jds..jde-1 with one halo row each side in memory -- a WRF patch, i.e.
    GLOBAL ids..jde but LOCAL jms:jme = jlo-1:jhi+1 and jts:jte = jlo:jhi.  The boundary
    flags then clip the first/last slab exactly as they clip the unsplit tile
    (module_small_step_em.f90:103-106)."""
    nrows = g.jde - g.jds            # computed rows jds .. jde-1
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    if nrows < world:
        raise ValueError(f"{nrows} rows cannot be split over {world} slabs")
    jlo = g.jds + (nrows * rank) // world
    jhi = g.jds + (nrows * (rank + 1)) // world - 1
    return g.replace(jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)


def patch_bounds(g: Bounds, ri: int, rj: int, pi: int, pj: int, *, align_elems: int = 1) -> Bounds:
    """Patch (ri, rj) of a pi x pj decomposition of the domain ``g`` in i and j (SURVEY.md section 8f
    row 4: domains whose j extent is too small for one slab per GPU).  Computed columns ids..ide-1
    and rows jds..jde-1 are split contiguously; memory holds the patch plus one halo cell on every
    side (``align_elems`` > 1 pads the i memory so that i = its starts an aligned boundary)."""
    ncol, nrow = g.ide - g.ids, g.jde - g.jds
    if not (0 <= ri < pi and 0 <= rj < pj) or ncol < pi or nrow < pj:
        raise ValueError("bad patch index or domain too small for this decomposition")
    ilo = g.ids + (ncol * ri) // pi
    ihi = g.ids + (ncol * (ri + 1)) // pi - 1
    jlo = g.jds + (nrow * rj) // pj
    jhi = g.jds + (nrow * (rj + 1)) // pj - 1
    if align_elems > 1:
        ims = ilo - align_elems
        idim = -(-(ihi + 1 - ims + 1) // align_elems) * align_elems
        ime = ims + idim - 1
    else:
        ims, ime = ilo - 1, ihi + 1
    return g.replace(ims=ims, ime=ime, its=ilo, ite=ihi, jms=jlo - 1, jme=jhi + 1, jts=jlo, jte=jhi)


@dataclass
class Patch:
    """One patch's arguments of advance_mu_t: bounds, flags, scalars and the 26 arrays
    (numpy on the host or torch tensors on a device)."""
    bounds: Bounds
    config: GridConfig
    arrays: dict
    rdx: float = RDX
    rdy: float = RDY
    dts: float = DTS
    epssm: float = EPSSM
    global_dims: tuple = _dc_field(default=())
    owner: object = None          # make_patch(native_domain=True): the amt_domain handle whose device arrays the tensors view

    def args(self):
        """The 48 arguments in the order of module_small_step_em.f90:7-18."""
        a = self.arrays
        return (a["ww"], a["ww_1"], a["u"], a["u_1"], a["v"], a["v_1"], a["mu"], a["mut"], a["muave"],
                a["muts"], a["muu"], a["muv"], a["mudf"], a["t"], a["t_1"], a["t_ave"], a["ft"],
                a["mu_tend"], self.rdx, self.rdy, self.dts, self.epssm, a["dnw"], a["fnm"], a["fnp"],
                a["rdnw"], a["msfuy"], a["msfvx_inv"], a["msftx"], a["msfty"], self.config,
                *self.bounds.as_tuple())

    def with_bounds(self, **kw) -> "Patch":
        return Patch(self.bounds.replace(**kw), self.config, self.arrays, self.rdx, self.rdy,
                     self.dts, self.epssm, self.global_dims, self.owner)

    def copy(self) -> "Patch":
        arrays = {k: (v.clone() if hasattr(v, "clone") else v.copy()) for k, v in self.arrays.items()}
        return Patch(self.bounds, self.config, arrays, self.rdx, self.rdy, self.dts, self.epssm, self.global_dims)

    def to_host(self) -> "Patch":
        arrays = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in self.arrays.items()}
        return Patch(self.bounds, self.config, arrays, self.rdx, self.rdy, self.dts, self.epssm, self.global_dims)

    def to_device(self, device="cuda") -> "Patch":
        import torch
        arrays = {k: torch.as_tensor(v).to(device) for k, v in self.arrays.items()}
        return Patch(self.bounds, self.config, arrays, self.rdx, self.rdy, self.dts, self.epssm, self.global_dims)


def _fill_args(b: Bounds, name: str, gdims):
    gni, gnk, gnj = gdims
    r = field_rank(name)
    idim, kdim, jdim = b.idim, b.kdim, b.jdim
    # global zero-based index of local element 0 = its Fortran index relative to the
    # minimal single-patch memory (0:NI+1, 1:NK+1, 0:NJ+1)
    return (idim, kdim, jdim, b.ims, b.kms - 1, b.jms, gni + 2, gnk + 1, gnj + 2), r


class NativeDomain:
    """Owner of a resident handle made by ``amt_domain_create`` (the C-ABI a Fortran or C host calls, placement sampling
    included) whose 26 device arrays torch tensors VIEW through ``__cuda_array_interface__``: a Python caller then holds
    exactly the memory a native host would.  Destroyed with the last tensor that views it."""

    def __init__(self, b: Bounds, config: GridConfig, itemsize: int):
        self.L = _lib.load_library()
        self.handle = ctypes.c_void_p()
        _lib.check(self.L.amt_domain_create(ctypes.byref(self.handle), itemsize, *config.as_ints(), *b.as_tuple()))

    def placement_ms(self):
        """Sweep time (ms) on every allocation of the state that amt_domain_create timed; [] when it did not sample."""
        ms = (ctypes.c_float * 16)()
        n = int(self.L.amt_domain_placement(self.handle, ms, 16))
        return [round(float(ms[k]), 3) for k in range(n) if ms[k] > 0]

    def view(self, field_id: int, shape, typestr: str):
        ptr = self.L.amt_domain_field_ptr(self.handle, field_id)
        if not ptr:
            raise _lib.AmtError(_lib.ERR_INVALID_ARG, "amt_domain_field_ptr returned NULL")

        class _View:                                   # keeps the owner alive for as long as a tensor views the array
            pass
        v = _View()
        v.owner = self
        v.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}
        return v

    def __del__(self):
        try:
            if self.handle:
                self.L.amt_domain_destroy(self.handle)
                self.handle = ctypes.c_void_p()
        except Exception:
            pass


def make_patch(b: Bounds, config: GridConfig = GridConfig(), dtype=np.float64, seed: int = 12345,
               global_dims=None, device=None, stream=None, native_domain: bool = False) -> Patch:
    """Fill all 26 arrays of a patch.  ``global_dims`` = (NI, NK, NJ) of the whole domain
    (default: derived from ``b`` assuming a single patch).  ``device=None`` -> numpy arrays
    via amt_synth_fill_host; otherwise torch tensors on ``device`` via amt_synth_fill_device.
    ``native_domain``: the device arrays are those of an ``amt_domain_create`` handle (``Patch.owner``) -- what a C / Fortran
    host holds, the library's default placement sampling included -- and the tensors view them."""
    L = _lib.load_library()
    if global_dims is None:
        global_dims = (b.ide - b.ids, b.kde - 1, b.jde - b.jds)
    arrays = {}
    if device is None:
        dt = np.dtype(dtype)
        for name in FIELD_NAMES:
            a = np.empty(b.shape(name), dtype=dt)
            fa, _ = _fill_args(b, name, global_dims)
            _lib.check(L.amt_synth_fill_host(FIELD_ID[name], dt.itemsize, a.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.c_uint64(seed), *fa))
            arrays[name] = a
    else:
        import torch
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)] \
            if not isinstance(dtype, torch.dtype) else dtype
        dev = torch.device(device)
        if stream is None:
            stream = torch.cuda.current_stream(dev)
        owner = None
        with torch.cuda.device(dev):
            if native_domain:
                owner = NativeDomain(b, config, 8 if tdt == torch.float64 else 4)
            for name in FIELD_NAMES:
                if owner is not None:
                    a = torch.as_tensor(owner.view(FIELD_ID[name], b.shape(name), "<f8" if tdt == torch.float64 else "<f4"), device=dev)
                    assert a.dtype == tdt and tuple(a.shape) == tuple(b.shape(name))
                else:
                    a = torch.empty(b.shape(name), dtype=tdt, device=dev)
                fa, _ = _fill_args(b, name, global_dims)
                _lib.check(L.amt_synth_fill_device(ctypes.c_void_p(stream.cuda_stream), FIELD_ID[name],
                                                   a.element_size(), ctypes.c_void_p(a.data_ptr()),
                                                   ctypes.c_uint64(seed), *fa))
                arrays[name] = a
        return Patch(b, config, arrays, RDX, RDY, DTS, EPSSM, tuple(global_dims), owner)
    return Patch(b, config, arrays, RDX, RDY, DTS, EPSSM, tuple(global_dims))


# ---------------------------------------------------------------------------------------------
# Inputs that change every sub-step.  In WRF advance_uv rewrites u and v before every advance_mu_t
# call, and the fields a patch sends to its neighbours (module_small_step_em.f90:143-146, 241-245)
# are what that call reads across the boundary: freshness is the point of a per-sub-step exchange.
# A static benchmark keeps them constant, and then an exchange that delivers ONCE satisfies every
# later sweep.  These helpers give sweep s (0-based) its own inputs -- the exchanged fields refilled
# from the generator with seed + s -- on the host (the checker's unsplit run) and on a device patch.
# ---------------------------------------------------------------------------------------------
EXCHANGED_INPUTS = ("u", "u_1", "v", "v_1", "t_1", "muu", "muv", "msfuy", "msfvx_inv")   # AMT_EXCHANGED_FIELDS
HALO_FROM_RIGHT = ("u", "u_1", "t_1", "muu", "msfuy")                                   # column i+1 is read (:145, :244)
HALO_FROM_LEFT = ("t_1",)                                                               # column i-1 is read (:245)
SIDE_BELOW, SIDE_ABOVE, SIDE_LEFT, SIDE_RIGHT = 1, 2, 4, 8                              # enum amt_sides


def sweep_seed(seed: int, sweep: int) -> int:
    """Seed of the exchanged inputs of sweep ``sweep`` (0-based; sweep 0 = the patch as made)."""
    return int(seed) + int(sweep)


def refresh_exchanged_inputs(patch: Patch, seed: int, sweep: int, stream=None) -> None:
    """Refill EXCHANGED_INPUTS of ``patch`` (whole local arrays, halos included) for sweep ``sweep``.
    Host patches: amt_synth_fill_host; device patches: amt_synth_fill_device on ``stream`` (default: torch's current)."""
    L = _lib.load_library()
    b = patch.bounds
    gdims = patch.global_dims or (b.ide - b.ids, b.kde - 1, b.jde - b.jds)
    s = ctypes.c_uint64(sweep_seed(seed, sweep))
    for name in EXCHANGED_INPUTS:
        a = patch.arrays[name]
        fa, _ = _fill_args(b, name, gdims)
        if hasattr(a, "is_cuda"):
            if a.is_cuda:
                import torch
                st = stream if stream is not None else torch.cuda.current_stream(a.device)
                with torch.cuda.device(a.device):
                    _lib.check(L.amt_synth_fill_device(ctypes.c_void_p(st.cuda_stream), FIELD_ID[name], a.element_size(),
                                                       ctypes.c_void_p(a.data_ptr()), s, *fa))
            else:                                         # a CPU tensor (the gloo tests): fill the memory it views
                _lib.check(L.amt_synth_fill_host(FIELD_ID[name], a.element_size(), ctypes.c_void_p(a.data_ptr()), s, *fa))
        else:
            _lib.check(L.amt_synth_fill_host(FIELD_ID[name], a.dtype.itemsize, a.ctypes.data_as(ctypes.c_void_p), s, *fa))


def poison_halos(patch: Patch, sides: int) -> None:
    """NaN into exactly what the stencil reads from a neighbour on ``sides`` (SIDE_* or-ed): row jte+1 of HALO_FROM_ABOVE,
    row jts-1 of t_1, column ite+1 of HALO_FROM_RIGHT, column its-1 of t_1.  Works on numpy arrays and torch tensors
    (device tensors: on torch's current stream)."""
    b, a, nan = patch.bounds, patch.arrays, float("nan")

    def fill(view):
        view.fill_(nan) if hasattr(view, "fill_") else view.fill(nan)

    if sides & SIDE_ABOVE:
        for n in HALO_FROM_ABOVE:
            fill(a[n][b.jte + 1 - b.jms])
    if sides & SIDE_BELOW:
        for n in HALO_FROM_BELOW:
            fill(a[n][b.jts - 1 - b.jms])
    if sides & SIDE_RIGHT:
        for n in HALO_FROM_RIGHT:
            fill(a[n][..., b.ite + 1 - b.ims])
    if sides & SIDE_LEFT:
        for n in HALO_FROM_LEFT:
            fill(a[n][..., b.its - 1 - b.ims])


def neighbour_sides(ri: int, rj: int, pi: int, pj: int) -> int:
    """The sides of patch (ri, rj) of pi x pj that have a neighbour."""
    return ((SIDE_BELOW if rj > 0 else 0) | (SIDE_ABOVE if rj < pj - 1 else 0)
            | (SIDE_LEFT if ri > 0 else 0) | (SIDE_RIGHT if ri < pi - 1 else 0))
