"""j-slab decomposition of advance_mu_t with a one-row input-halo exchange.

The reference splits the domain statically on j across its GPUs and refills three
redundant halo rows per side from the host copy on every call, with no inter-GPU
communication (advance_mu_t_no_async.cu:108-162).  Here one process owns one GPU and one
contiguous j-slab (``synth.slab_bounds``), keeps it resident, and before each sweep trades
exactly the rows the stencil reads across a slab edge:

  * from the slab above (j+1):  row jhi+1 of  v, v_1, t_1 (3-D)  and  muv, msfvx_inv (2-D)
    (module_small_step_em.f90:143, :241)
  * from the slab below (j-1):  row jlo-1 of  t_1                                  (:242)

All of them are pure inputs, so the exchange runs on a second stream while the interior
rows jlo+1..jhi-1 are computed; only the two edge rows wait for it.  In (i,k,j) layout a
j-row is one contiguous run of idim*kdim elements, so rows are sent in place (no packing).
Transport: ``torch.distributed`` point-to-point -- backend "nccl" is RCCL send/recv over
xGMI on ROCm; the same code runs over "gloo" on CPU tensors in the tests.  No collective
is needed anywhere in this path.
"""
from __future__ import annotations

from typing import Callable, Optional

from . import synth as _S
from .synth import Patch, HALO_FROM_ABOVE, HALO_FROM_BELOW


def _dist():
    import torch.distributed as dist
    return dist


def _fault(step: str, occurrence: int) -> bool:
    """AMT_TEST_FAULT="<step>@<n>" (csrc/amt_internal.h): the n-th exchange of a torch stepper is skipped by every rank
    ("skip_exchange") -- the halo-freshness tests must turn red on it.  Unset outside those tests."""
    import os
    spec = os.environ.get("AMT_TEST_FAULT", "")
    return bool(spec) and spec == f"{step}@{occurrence}"


class SlabStepper:
    """Runs advance_mu_t sweeps on one j-slab of a domain split over ``world`` ranks.

    ``patch``   : this rank's Patch (bounds from ``synth.slab_bounds``; arrays are torch
                  tensors -- CUDA for the product path, CPU in the gloo tests).
    ``compute`` : callable(*the 48 advance_mu_t arguments, stream=...) that updates a tile in
                  place.  The product passes ``api.advance_mu_t`` (HIP); there is no default
                  CPU implementation.
    """

    def __init__(self, patch: Patch, rank: int, world: int, compute: Callable, *,
                 group=None, overlap: bool = True, variant: int = 0, transport: Optional[Callable] = None,
                 stage_through_host: bool = False):
        self.patch, self.rank, self.world = patch, rank, world
        self.compute, self.group, self.overlap, self.variant = compute, group, overlap, variant
        # transport(stepper): replaces the torch.distributed exchange (tests run several slabs of
        # one domain in ONE process on one GPU and copy the halo rows device-to-device)
        self.transport = transport
        self._bound = {}          # (jts, jte) -> pre-marshalled device call (see api.bind_device_call)
        # stage_through_host: the process group cannot move device memory (gloo): bounce every row
        # through a host buffer.  Debug/bring-up only -- lets the whole N > 1 bench path run with
        # several ranks sharing one GPU, which RCCL refuses.
        self.stage_through_host = stage_through_host
        self.below: Optional[int] = rank - 1 if rank > 0 else None
        self.above: Optional[int] = rank + 1 if rank < world - 1 else None
        any_arr = patch.arrays["t_1"]
        self.on_gpu = bool(getattr(any_arr, "is_cuda", False))
        self.main_stream = self.comm_stream = None
        if self.on_gpu:
            import torch
            self.main_stream = torch.cuda.current_stream(any_arr.device)
            # high priority: the exchange and the edge rows are small and the sweep's join waits for them;
            # they must not queue behind the interior's remaining rounds of workgroups
            self.comm_stream = torch.cuda.Stream(device=any_arr.device, priority=-1)
        b = patch.bounds
        if world > 1 and (b.jme - b.jms + 1) != (b.jte - b.jts + 1) + 2:
            raise ValueError("a slab patch holds its rows plus exactly one halo row per side")

    # -- halo exchange -----------------------------------------------------------------
    def _p2p_ops(self):
        dist = _dist()
        a = self.patch.arrays
        jdim = self.patch.bounds.jdim
        first_owned, last_owned, halo_lo, halo_hi = 1, jdim - 2, 0, jdim - 1
        ops = []
        # order per peer pair is fixed (NCCL/RCCL matches send/recv by order, gloo by tag)
        if self.below is not None:
            for n, name in enumerate(HALO_FROM_ABOVE):     # my first row is their row jhi+1
                ops.append(dist.P2POp(dist.isend, a[name][first_owned], self.below, self.group, tag=10 + n))
            for n, name in enumerate(HALO_FROM_BELOW):
                ops.append(dist.P2POp(dist.irecv, a[name][halo_lo], self.below, self.group, tag=20 + n))
        if self.above is not None:
            for n, name in enumerate(HALO_FROM_ABOVE):
                ops.append(dist.P2POp(dist.irecv, a[name][halo_hi], self.above, self.group, tag=10 + n))
            for n, name in enumerate(HALO_FROM_BELOW):     # my last row is their row jlo-1
                ops.append(dist.P2POp(dist.isend, a[name][last_owned], self.above, self.group, tag=20 + n))
        return ops

    def exchange_halos(self):
        """Post the sends/receives of one sweep and wait for them on the current stream
        (device-side wait for RCCL; host wait for gloo)."""
        self._exchanges = getattr(self, "_exchanges", 0) + 1
        if _fault("skip_exchange", self._exchanges):
            return
        if self.transport is not None:
            self.transport(self)
            return
        ops = self._p2p_ops()
        if not ops:
            return
        if self.stage_through_host and self.on_gpu:
            dist = _dist()
            staged, recvs = [], []
            for op in ops:
                if op.op is dist.isend:
                    staged.append(dist.P2POp(dist.isend, op.tensor.cpu(), op.peer, op.group, op.tag))
                else:
                    host = op.tensor.new_empty(op.tensor.shape, device="cpu")
                    recvs.append((op.tensor, host))
                    staged.append(dist.P2POp(dist.irecv, host, op.peer, op.group, op.tag))
            for req in dist.batch_isend_irecv(staged):
                req.wait()
            for dev, host in recvs:
                dev.copy_(host, non_blocking=False)
            return
        for req in _dist().batch_isend_irecv(ops):
            req.wait()

    def halo_bytes_per_sweep(self) -> int:
        """Bytes this rank sends + receives per sweep."""
        a = self.patch.arrays
        n = 0
        for peer, names in ((self.below, HALO_FROM_ABOVE + HALO_FROM_BELOW),
                            (self.above, HALO_FROM_ABOVE + HALO_FROM_BELOW)):
            if peer is not None:
                n += sum(a[name][0].numel() * a[name].element_size() for name in names)
        return n

    # -- one sweep ---------------------------------------------------------------------
    def _tile(self, jts: int, jte: int, stream, beside: bool = False):
        if jte < jts:
            return
        if self.on_gpu:
            variant = self.variant | (0x100 if beside else 0)        # LAUNCH_BESIDE_OTHERS: leave round boundaries to the exchange
            key = (jts, jte, id(stream), beside)
            call = self._bound.get(key)
            if call is None:
                args = self.patch.with_bounds(jts=jts, jte=jte).args()
                binder = getattr(self.compute, "bind", None)
                if binder is None:                   # a plain callable: marshal on every call
                    self.compute(*args, stream=stream, variant=variant)
                    return
                call = self._bound[key] = binder(*args, stream=stream, variant=variant)
            call()
        else:
            self.compute(*self.patch.with_bounds(jts=jts, jte=jte).args())

    def step(self):
        b = self.patch.bounds
        jlo, jhi = b.jts, b.jte
        if self.world == 1 or (self.below is None and self.above is None):
            self._tile(jlo, jhi, self.main_stream)
            return
        # rows that read a neighbour's data: jlo (if there is a slab below), jhi (if above)
        in_lo = jlo + (1 if self.below is not None else 0)
        in_hi = jhi - (1 if self.above is not None else 0)
        if self.on_gpu and self.overlap:
            import torch
            # comm stream: exchange, then the edge rows (they need nothing but the halos and the
            # inputs).  main stream: the interior.  The small edge launches (one row each) then fill
            # the CUs that the interior's last, partly filled round of workgroups leaves idle.
            self.comm_stream.wait_stream(self.main_stream)      # inputs of this sub-step are final
            self._tile(in_lo, in_hi, self.main_stream, beside=True)   # interior overlaps the exchange
            with torch.cuda.stream(self.comm_stream):
                self.exchange_halos()                           # RCCL send/recv on the comm stream
                self._edges(jlo, jhi, self.comm_stream)
            self.main_stream.wait_stream(self.comm_stream)
        else:
            self.exchange_halos()
            self._tile(in_lo, in_hi, self.main_stream)
            self._edges(jlo, jhi, self.main_stream)

    def _edges(self, jlo, jhi, stream):
        if self.below is not None:
            self._tile(jlo, min(jlo, jhi), stream)
        if self.above is not None and (jhi > jlo or self.below is None):
            self._tile(jhi, jhi, stream)


# i-direction halos (module_small_step_em.f90:145 reads u, u_1, muu, msfuy at i+1; :244-245 read
# t_1 at i+1 and i-1)
from .synth import HALO_FROM_RIGHT, HALO_FROM_LEFT  # noqa: E402  (re-exported: tests name them through this module)


class GridStepper:
    """advance_mu_t on patch (ri, rj) of a pi x pj decomposition in i AND j (SURVEY.md section 8f
    row 4).  j halos are contiguous rows and travel in place as in SlabStepper; i halos are columns
    (stride idim in memory), so they are packed into contiguous buffers, sent, and unpacked on
    arrival.  No diagonal neighbours are needed: the stencil reads (i+-1, j) and (i, j+-1) only.
    One launch per sweep after the exchange (no interior/edge split here: this decomposition is for
    domains too small in j to be bandwidth-bound per GPU).  Ranks are row-major: rank = rj*pi + ri.
    """

    def __init__(self, patch: Patch, ri: int, rj: int, pi: int, pj: int, compute: Callable, *,
                 group=None, variant: int = 0, stage_through_host: bool = False,
                 native: Optional[str] = None, unique_id: Optional[bytes] = None, overlap: bool = True):
        self.patch, self.ri, self.rj, self.pi, self.pj = patch, ri, rj, pi, pj
        self.compute, self.group, self.variant = compute, group, variant
        self.stage_through_host = stage_through_host
        # native="rccl" | "ipc": a device patch is stepped by the C++ runtime (amt_grid_*: HIP pack / unpack kernels, one
        # exchange, interior beside it) and this class only forwards; None: the torch.distributed path below (CPU tensors
        # over gloo in the tests; device tensors staged through the host for bring-up)
        self._native = None
        if native is not None:
            self._native = NativeGridStepper(patch, ri, rj, pi, pj, unique_id, overlap=overlap, variant=variant, transport=native)
        rank = lambda i, j: j * pi + i
        self.left = rank(ri - 1, rj) if ri > 0 else None
        self.right = rank(ri + 1, rj) if ri < pi - 1 else None
        self.below = rank(ri, rj - 1) if rj > 0 else None
        self.above = rank(ri, rj + 1) if rj < pj - 1 else None
        b = patch.bounds
        self.c_first, self.c_last = b.its - b.ims, b.ite - b.ims          # owned columns (memory index)
        self.c_halo_l, self.c_halo_r = self.c_first - 1, self.c_last + 1
        self.on_gpu = bool(getattr(patch.arrays["t_1"], "is_cuda", False))
        self._call = None

    def _col(self, name, c):
        a = self.patch.arrays[name]
        return a[..., c]                                   # (jdim, kdim) or (jdim,) strided view

    def exchange_halos(self):
        if self._native is not None:
            return self._native.exchange_halos()
        self._exchanges = getattr(self, "_exchanges", 0) + 1
        if _fault("skip_exchange", self._exchanges):
            return
        dist = _dist()
        a = self.patch.arrays
        jdim = self.patch.bounds.jdim
        ops, unpack = [], []
        host = self.stage_through_host and self.on_gpu

        def send(t, peer, tag):
            t = t.contiguous()
            ops.append(dist.P2POp(dist.isend, t.cpu() if host else t, peer, self.group, tag))

        def recv(view, peer, tag, packed):
            if packed or host:
                buf = view.new_empty(view.shape, device="cpu" if host else view.device)
                unpack.append((view, buf))
                ops.append(dist.P2POp(dist.irecv, buf, peer, self.group, tag))
            else:
                ops.append(dist.P2POp(dist.irecv, view, peer, self.group, tag))

        if self.below is not None:
            for n, name in enumerate(HALO_FROM_ABOVE):
                send(a[name][1], self.below, 10 + n)
            recv(a["t_1"][0], self.below, 20, False)
        if self.above is not None:
            for n, name in enumerate(HALO_FROM_ABOVE):
                recv(a[name][jdim - 1], self.above, 10 + n, False)
            send(a["t_1"][jdim - 2], self.above, 20)
        if self.left is not None:
            for n, name in enumerate(HALO_FROM_RIGHT):     # my first column is their column ihi+1
                send(self._col(name, self.c_first), self.left, 30 + n)
            recv(self._col("t_1", self.c_halo_l), self.left, 40, True)
        if self.right is not None:
            for n, name in enumerate(HALO_FROM_RIGHT):
                recv(self._col(name, self.c_halo_r), self.right, 30 + n, True)
            send(self._col("t_1", self.c_last), self.right, 40)
        if not ops:
            return
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for view, buf in unpack:
            view.copy_(buf)

    def step(self):
        if self._native is not None:
            return self._native.step(1)
        self.exchange_halos()
        args = self.patch.args()
        if self.on_gpu:
            if self._call is None:
                binder = getattr(self.compute, "bind", None)
                if binder is None:
                    self.compute(*args, variant=self.variant)
                    return
                self._call = binder(*args, variant=self.variant)
            self._call()
        else:
            self.compute(*args)


def _substep_inputs(stepper, dom, seed, sweep, sides):
    """amt_domain_fill_fields(AMT_EXCHANGED_FIELDS, seed + sweep) [+ amt_domain_poison_halos(sides)] on the domain's stream."""
    b = stepper.patch.bounds
    gdims = stepper.patch.global_dims or (b.ide - b.ids, b.kde - 1, b.jde - b.jds)
    mask = 0
    for name in _S.EXCHANGED_INPUTS:
        mask |= 1 << _S.FIELD_ID[name]
    with stepper._dev():
        stepper._lib.check(stepper.L.amt_domain_fill_fields(dom, mask, _S.sweep_seed(seed, sweep), b.ims, b.kms - 1, b.jms,
                                                            gdims[0] + 2, gdims[1] + 1, gdims[2] + 2))
        if sides:
            stepper._lib.check(stepper.L.amt_domain_poison_halos(dom, sides))


class NativeSlabStepper:
    """The same j-slab sweep as ``SlabStepper`` driven by the C++ runtime behind the C-ABI
    (``amt_slab_*``, include/amt_advance_mu_t.h section 5): ``ncclSend/ncclRecv`` of the halo rows
    in one RCCL group on a communication stream, the two edge rows behind them on that stream, the
    interior on the patch's stream.  This is the path a Fortran or C host calls
    (fortran/advance_mu_t_slab_driver.f90); Python only hands over pointers.

    ``patch`` holds torch CUDA tensors (they stay the owners: ``amt_domain_wrap``); ``stream`` is
    the torch stream the sweeps are enqueued on; ``unique_id`` are the AMT_UNIQUE_ID_BYTES every rank
    got from rank 0's ``comm_unique_id()`` (None when world == 1 and not loopback).
    ``transport``: "rccl" (ncclSend/ncclRecv), or "ipc" (peer copies between the processes of one node through
    hipIpcMemHandles and a shared-memory mailbox: no RCCL, ranks may share one device).
    Creating it is collective over the ``world`` ranks (ncclCommInitRank / the IPC set-up).
    """

    NO_OVERLAP, LOOPBACK, TRANSPORT_IPC = 1, 2, 4          # enum amt_slab_flags

    def __init__(self, patch: Patch, rank: int, world: int, unique_id: Optional[bytes] = None, *,
                 stream=None, overlap: bool = True, variant: int = 0, loopback: bool = False, transport: str = "rccl"):
        import ctypes
        import torch
        from . import lib as _lib
        from .synth import FIELD_NAMES
        self._lib, self._ct = _lib, ctypes
        self.L = L = _lib.load_library()
        self.patch, self.rank, self.world = patch, rank, world
        self.below: Optional[int] = rank - 1 if rank > 0 else None
        self.above: Optional[int] = rank + 1 if rank < world - 1 else None
        self._halo_sides = (_S.SIDE_BELOW | _S.SIDE_ABOVE) if loopback else _S.neighbour_sides(0, rank, 1, world)
        t0 = patch.arrays["t_1"]
        if not t0.is_cuda:
            raise TypeError("NativeSlabStepper needs a device patch (there is no CPU path)")
        b = patch.bounds
        for name in FIELD_NAMES:
            t = patch.arrays[name]
            if not (t.is_cuda and t.dtype == t0.dtype and t.is_contiguous() and tuple(t.shape) == tuple(b.shape(name))):
                raise TypeError(f"{name}: need a contiguous device tensor of shape {b.shape(name)}")
        self.stream = stream if stream is not None else torch.cuda.Stream(device=t0.device)
        self.device_index = t0.device.index if t0.device.index is not None else torch.cuda.current_device()
        fields = (ctypes.c_void_p * len(FIELD_NAMES))(*[patch.arrays[n].data_ptr() for n in FIELD_NAMES])
        self._dom, self._slab = ctypes.c_void_p(), ctypes.c_void_p()
        with torch.cuda.device(self.device_index):
            _lib.check(L.amt_domain_wrap(ctypes.byref(self._dom), t0.element_size(), *patch.config.as_ints(),
                                         *b.as_tuple(), fields, ctypes.c_void_p(self.stream.cuda_stream)))
            try:
                _lib.check(L.amt_domain_set_scalars(self._dom, patch.rdx, patch.rdy, patch.dts, patch.epssm))
                _lib.check(L.amt_domain_set_variant(self._dom, int(variant)))
                if transport not in ("rccl", "ipc"):
                    raise ValueError("transport is 'rccl' or 'ipc'")
                flags = ((0 if overlap else self.NO_OVERLAP) | (self.LOOPBACK if loopback else 0)
                         | (self.TRANSPORT_IPC if transport == "ipc" else 0))
                uid = None
                if unique_id is not None:
                    uid = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id))
                _lib.check(L.amt_slab_create(ctypes.byref(self._slab), self._dom, rank, world, uid, flags))
            except BaseException:
                L.amt_domain_destroy(self._dom)
                self._dom = ctypes.c_void_p()
                raise

    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: a fresh communicator id to hand to every rank (ncclGetUniqueId)."""
        import ctypes
        from . import lib as _lib
        uid = (ctypes.c_char * 128)()
        _lib.check(_lib.load_library().amt_comm_unique_id(uid))
        return bytes(uid)

    def _dev(self):
        import torch
        return torch.cuda.device(self.device_index)

    def step(self, n_sweeps: int = 1):
        with self._dev():
            self._lib.check(self.L.amt_slab_step(self._slab, int(n_sweeps)))

    def exchange_halos(self):
        with self._dev():
            self._lib.check(self.L.amt_slab_exchange(self._slab))

    def sync(self):
        with self._dev():
            self._lib.check(self.L.amt_slab_sync(self._slab))

    def next_substep_inputs(self, seed: int, sweep: int, poison: bool = True):
        """What a host model does between two calls: new values in the fields that cross a slab boundary (the stand-in for
        advance_uv: amt_domain_fill_fields with AMT_EXCHANGED_FIELDS, seed + sweep) and, for verification, NaN in the halo
        rows (amt_domain_poison_halos) -- both on the domain's stream.  Only an exchange that delivers THIS sweep's rows
        then gives the bits of the unsplit run."""
        _substep_inputs(self, self._dom, seed, sweep, self._halo_sides if poison else 0)

    def set_skew_us(self, microseconds: int):
        """Test hook: every later sweep's exchange starts this late on the communication stream (neighbour skew)."""
        self._lib.check(self.L.amt_slab_set_skew_us(self._slab, int(microseconds)))

    def halo_bytes_per_sweep(self) -> int:
        return int(self.L.amt_slab_halo_bytes(self._slab))           # sent + received

    def transport(self) -> str:
        """'rccl', 'ipc' or 'none' -- what carries this stepper's halo rows (amt_slab_transport)."""
        return self.L.amt_slab_transport(self._slab).decode()

    def pull_mode(self) -> str:
        """IPC transport: 'copy engine' or 'fused kernel' (amt_slab_pull_mode); '' with RCCL."""
        return self.L.amt_slab_pull_mode(self._slab).decode()

    def comm_info(self):
        """(rank, world) as the transport reports them (communicator / ranks attached to the IPC block); (0, 1) without one."""
        r, w = self._ct.c_int(), self._ct.c_int()
        self._lib.check(self.L.amt_slab_comm_info(self._slab, self._ct.byref(r), self._ct.byref(w)))
        return r.value, w.value

    def close(self):
        if self._slab:
            with self._dev():
                self.L.amt_slab_destroy(self._slab)
            self._slab = self._ct.c_void_p()
        if self._dom:
            self.L.amt_domain_destroy(self._dom)
            self._dom = self._ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeGridStepper:
    """advance_mu_t on patch (ri, rj) of pi x pj driven by the C++ runtime behind the C-ABI (``amt_grid_*``, header section 5b):
    HIP pack / unpack kernels for the strided halo columns, rows and packed columns in ONE exchange (RCCL group or IPC
    pulls), interior cells on the patch's stream beside it, boundary rows and columns behind it on the communication stream.
    Same construction as ``NativeSlabStepper`` (which is its pi = 1 case); rank = rj * pi + ri.
    """

    NO_OVERLAP, LOOPBACK, TRANSPORT_IPC = 1, 2, 4          # enum amt_slab_flags

    def __init__(self, patch: Patch, ri: int, rj: int, pi: int, pj: int, unique_id: Optional[bytes] = None, *,
                 stream=None, overlap: bool = True, variant: int = 0, loopback: bool = False, transport: str = "rccl"):
        import ctypes
        import torch
        from . import lib as _lib
        from .synth import FIELD_NAMES
        self._lib, self._ct = _lib, ctypes
        self.L = L = _lib.load_library()
        self.patch, self.ri, self.rj, self.pi, self.pj = patch, ri, rj, pi, pj
        self._halo_sides = 15 if loopback else _S.neighbour_sides(ri, rj, pi, pj)
        t0 = patch.arrays["t_1"]
        if not t0.is_cuda:
            raise TypeError("NativeGridStepper needs a device patch (there is no CPU path)")
        if transport not in ("rccl", "ipc"):
            raise ValueError("transport is 'rccl' or 'ipc'")
        b = patch.bounds
        for name in FIELD_NAMES:
            t = patch.arrays[name]
            if not (t.is_cuda and t.dtype == t0.dtype and t.is_contiguous() and tuple(t.shape) == tuple(b.shape(name))):
                raise TypeError(f"{name}: need a contiguous device tensor of shape {b.shape(name)}")
        self.stream = stream if stream is not None else torch.cuda.Stream(device=t0.device)
        self.device_index = t0.device.index if t0.device.index is not None else torch.cuda.current_device()
        fields = (ctypes.c_void_p * len(FIELD_NAMES))(*[patch.arrays[n].data_ptr() for n in FIELD_NAMES])
        self._dom, self._grid = ctypes.c_void_p(), ctypes.c_void_p()
        with torch.cuda.device(self.device_index):
            _lib.check(L.amt_domain_wrap(ctypes.byref(self._dom), t0.element_size(), *patch.config.as_ints(),
                                         *b.as_tuple(), fields, ctypes.c_void_p(self.stream.cuda_stream)))
            try:
                _lib.check(L.amt_domain_set_scalars(self._dom, patch.rdx, patch.rdy, patch.dts, patch.epssm))
                _lib.check(L.amt_domain_set_variant(self._dom, int(variant)))
                flags = ((0 if overlap else self.NO_OVERLAP) | (self.LOOPBACK if loopback else 0)
                         | (self.TRANSPORT_IPC if transport == "ipc" else 0))
                uid = None
                if unique_id is not None:
                    uid = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id))
                _lib.check(L.amt_grid_create(ctypes.byref(self._grid), self._dom, ri, rj, pi, pj, uid, flags))
            except BaseException:
                L.amt_domain_destroy(self._dom)
                self._dom = ctypes.c_void_p()
                raise

    comm_unique_id = staticmethod(NativeSlabStepper.comm_unique_id)

    def _dev(self):
        import torch
        return torch.cuda.device(self.device_index)

    def step(self, n_sweeps: int = 1):
        with self._dev():
            self._lib.check(self.L.amt_grid_step(self._grid, int(n_sweeps)))

    def exchange_halos(self):
        with self._dev():
            self._lib.check(self.L.amt_grid_exchange(self._grid))

    def sync(self):
        with self._dev():
            self._lib.check(self.L.amt_grid_sync(self._grid))

    def next_substep_inputs(self, seed: int, sweep: int, poison: bool = True):
        """As NativeSlabStepper.next_substep_inputs, for the halo rows AND columns of a patch."""
        _substep_inputs(self, self._dom, seed, sweep, self._halo_sides if poison else 0)

    def halo_bytes_per_sweep(self) -> int:
        return int(self.L.amt_grid_halo_bytes(self._grid))           # sent + received

    def transport(self) -> str:
        return self.L.amt_grid_transport(self._grid).decode()

    def pull_mode(self) -> str:
        return self.L.amt_grid_pull_mode(self._grid).decode()

    def comm_info(self):
        r, w = self._ct.c_int(), self._ct.c_int()
        self._lib.check(self.L.amt_grid_comm_info(self._grid, self._ct.byref(r), self._ct.byref(w)))
        return r.value, w.value

    def close(self):
        if self._grid:
            with self._dev():
                self.L.amt_grid_destroy(self._grid)
            self._grid = self._ct.c_void_p()
        if self._dom:
            self.L.amt_domain_destroy(self._dom)
            self._dom = self._ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
