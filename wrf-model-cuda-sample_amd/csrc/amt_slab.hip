// amt_slab.hip -- j-slab stepping with one-row halos, amt_slab_* (include/amt_advance_mu_t.h section 5): the native twin
// of patch.SlabStepper, for C / Fortran hosts that run one process per GPU (the reference splits j
// over its GPUs inside one process with host-sourced halos, advance_mu_t_no_async.cu:108-162).  The rows travel through the
// exchange engine of amt_comm.h: RCCL send/recv, or peer copies between processes (AMT_SLAB_TRANSPORT_IPC).
#include "amt_comm.h"
#include <vector>

struct amt_slab {
    amt_domain *dom = nullptr;
    int rank = 0, world = 1;
    int below = -1, above = -1;          // neighbour ranks, -1 = none
    bool overlap = true;
    AmtExchange *xchg = nullptr;         // the rows that cross the slab's two boundaries, and the transport under them
    hipStream_t comm_stream = nullptr;
    hipEvent_t inputs_final = nullptr, edges_done = nullptr, t0 = nullptr, t1 = nullptr;
    int skew_us = 0;                     // test hook: the neighbours' rows arrive this late (amt_slab_set_skew_us)
};

namespace {
// rows that cross a slab boundary: row jte+1 of these comes from the rank above (its row jts) ...
const int kHaloFromAbove[] = {AMT_F_V, AMT_F_V_1, AMT_F_T_1, AMT_F_MUV, AMT_F_MSFVX_INV};   // :143-144, :241
// ... and row jts-1 of t_1 from the rank below (its row jte), :242
const int kHaloFromBelow[] = {AMT_F_T_1};
}  // namespace

extern "C" int amt_slab_destroy(amt_slab *s)
{
    if (!s) return AMT_OK;
    DeviceScope scope(s->dom ? s->dom->device : 0);
    if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
    (void)amt_exchange_destroy(s->xchg);
    for (hipEvent_t e : {s->inputs_final, s->edges_done, s->t0, s->t1})
        if (e) (void)hipEventDestroy(e);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    delete s;
    return AMT_OK;
}

extern "C" int amt_slab_create(amt_slab **out, amt_domain *dom, int rank, int world, const void *unique_id, int flags)
{
    if (!out) return amt_fail(AMT_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (!dom || world < 1 || rank < 0 || rank >= world) return amt_fail(AMT_ERR_INVALID_ARG, "bad slab argument");
    const bool loopback = (flags & AMT_SLAB_LOOPBACK) != 0;
    if (loopback && world != 1) return amt_fail(AMT_ERR_INVALID_ARG, "AMT_SLAB_LOOPBACK is a one-rank test mode");
    const bool comm_needed = world > 1 || loopback;
    if (comm_needed && !unique_id) return amt_fail(AMT_ERR_INVALID_ARG, "a communicator needs the unique id");
    if (comm_needed && (dom->jts - 1 < dom->jms || dom->jte + 1 > dom->jme))
        return amt_fail(AMT_ERR_PRECONDITION, "a slab holds one halo row below jts and above jte");
    int transport = (flags & AMT_SLAB_TRANSPORT_IPC) ? AMT_XCHG_IPC : AMT_XCHG_RCCL;
    if (const char *e = getenv("AMT_SLAB_TRANSPORT")) {                  // hosts that cannot pass the flag (the Fortran drivers)
        if (!strcmp(e, "ipc")) transport = AMT_XCHG_IPC;
        else if (!strcmp(e, "rccl")) transport = AMT_XCHG_RCCL;
        else if (*e) return amt_fail(AMT_ERR_INVALID_ARG, "AMT_SLAB_TRANSPORT must be rccl or ipc, not '%s'", e);
    }
    amt_slab *s = new (std::nothrow) amt_slab;
    if (!s) return amt_fail(AMT_ERR_ALLOC, "host allocation failed");
    s->dom = dom; s->rank = rank; s->world = world;
    s->overlap = !(flags & AMT_SLAB_NO_OVERLAP);
    s->below = loopback ? rank : rank > 0 ? rank - 1 : -1;
    s->above = loopback ? rank : rank < world - 1 ? rank + 1 : -1;
    DeviceScope scope(dom->device);
    // the communication stream outranks the domain's: the exchange and the two edge rows behind it are
    // small and the sweep's join waits for them, so they must not queue behind the interior's
    // remaining rounds of workgroups
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    hipError_t e = hipStreamCreateWithPriority(&s->comm_stream, hipStreamNonBlocking, prio_high);
    for (hipEvent_t *ev : {&s->inputs_final, &s->edges_done})
        if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    for (hipEvent_t *ev : {&s->t0, &s->t1})
        if (e == hipSuccess) e = hipEventCreate(ev);
    if (e != hipSuccess) {
        amt_slab_destroy(s);
        return amt_fail(AMT_ERR_HIP, "amt_slab_create: %s", hipGetErrorString(e));
    }
    // the rows of one exchange, in place (a j row of the (i,k,j) layout is one contiguous run).  Per pair of ranks the
    // order of the sends is the order of the receives on the other side.
    const size_t idim = dom->ime - dom->ims + 1, kdim = dom->kme - dom->kms + 1;
    auto row = [&](int f, int j, int peer) {
        const size_t count = amt_field_rank(f) == 3 ? idim * kdim : idim;
        return AmtSeg{static_cast<char *>(dom->field[f]) + (size_t)(j - dom->jms) * count * dom->dtype_bytes, count * dom->dtype_bytes, peer};
    };
    std::vector<AmtSeg> sends, recvs;
    if (s->below >= 0) for (int f : kHaloFromAbove) sends.push_back(row(f, dom->jts, s->below));
    if (s->above >= 0) for (int f : kHaloFromBelow) sends.push_back(row(f, dom->jte, s->above));
    if (s->above >= 0) for (int f : kHaloFromAbove) recvs.push_back(row(f, dom->jte + 1, s->above));
    if (s->below >= 0) for (int f : kHaloFromBelow) recvs.push_back(row(f, dom->jts - 1, s->below));
    int rc = amt_exchange_create(&s->xchg, transport, rank, world, unique_id, dom->device, sends.data(), (int)sends.size(),
                                 recvs.data(), (int)recvs.size(), loopback);
    if (rc) {
        const std::string keep = amt_last_error();
        amt_slab_destroy(s);
        return amt_fail(rc, "%s", keep.c_str());
    }
    *out = s;
    return AMT_OK;
}

namespace {
// Test hook (amt_slab_set_skew_us): holds the communication stream for `ticks` of the 100 MHz real-time counter, so that
// the exchange behind it starts -- and the neighbours' rows arrive -- that much late: neighbour skew on one GPU.
// AMT_SLAB_SKEW_WGS=n (default 1) gives the delay the footprint of RCCL's waiting send/recv kernel: n workgroups of 256 threads,
// each with enough LDS to have a compute unit to itself (RCCL's 31 workgroups fit beside no march workgroup either).
__global__ void amt_slab_delay_kernel(unsigned long long ticks)
{
    extern __shared__ unsigned char amt_delay_lds[];
    if (threadIdx.x == 0) amt_delay_lds[0] = 0;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

template <typename T>
int amt_slab_tile(amt_slab *s, hipStream_t stream, int jts, int jte, bool beside_the_exchange = false)
{
    if (jte < jts) return AMT_OK;
    AmtArgs<T> a;
    amt_domain_args<T>(s->dom, a);
    a.jts = jts; a.jte = jte;
    // A launch that is ONE round of workgroups (the launcher's choice for a slab on its own) holds every compute unit
    // until it ends: the exchange's kernels and the edge rows would start behind it and the wire time would show in full
    // (profiles/r04_raw/slab_order/).  Beside the exchange the interior is planned in at least two rounds.
    return beside_the_exchange ? amt_device_call_shared<T>(stream, s->dom->variant, a) : amt_device_call<T>(stream, s->dom->variant, a);
}

template <typename T>
int amt_slab_step_t(amt_slab *s, int n_sweeps)
{
    amt_domain *d = s->dom;
    const int jlo = d->jts, jhi = d->jte;
    const bool lo = s->below >= 0, hi = s->above >= 0;
    // the rows the routine really updates in this slab (module_small_step_em.f90:91-106): with specified /
    // nested boundaries the first / last row of an outermost slab is clipped away.  The one-launch edge path
    // computes rows j_start and j_end of the CLIPPED window, so it is only right when nothing is clipped
    // (always, for a slab with a neighbour on that side -- except in loopback, where the rank is its own
    // neighbour on a slab that touches the domain edge).
    const AmtWindow wclip = amt_window(d->periodic_x, d->specified, d->nested, d->ids, d->ide, d->jds, d->jde,
                                       d->its, d->ite, jlo, jhi, d->kts, d->kte);
    const bool unclipped = wclip.j_start == jlo && wclip.j_end == jhi;
    // a failure between the fork (inputs_final) and the join (edges_done) must not leave the streams apart
    auto join = [&]() {
        if (s->overlap) {
            (void)hipEventRecord(s->edges_done, s->comm_stream);
            (void)hipStreamWaitEvent(d->stream, s->edges_done, 0);
        }
    };
    for (int sweep = 0; sweep < n_sweeps; ++sweep) {
        int rc = AMT_OK;
        if (!lo && !hi) {
            rc = amt_slab_tile<T>(s, d->stream, jlo, jhi);
            if (rc) return rc;
            continue;
        }
        // rows that read a neighbour's data: jlo (slab below), jhi (slab above); the rest is interior
        const int in_lo = jlo + (lo ? 1 : 0), in_hi = jhi - (hi ? 1 : 0);
        hipStream_t edge_stream = s->overlap ? s->comm_stream : d->stream;
        // Order of the enqueues (profiles/r05_slab_ab.md).  A march workgroup takes a compute unit whole, so whatever the
        // communication stream launches once the interior is out starts only where an interior workgroup ends.  With the IPC
        // transport the exchange therefore goes out FIRST: its waiting kernel has its compute unit(s) from the start of the
        // sweep and the rows are in as soon as the neighbour has them.  RCCL's send/recv kernel would hold 31 units for as
        // long as it waits, so there the interior keeps its head start (AMT_SLAB_EXCHANGE_FIRST=0|1 overrides either).
        static const int order_env = [] { const char *e = getenv("AMT_SLAB_EXCHANGE_FIRST"); return e && *e ? atoi(e) : -1; }();
        const bool exchange_first = order_env >= 0 ? order_env != 0 : amt_exchange_transport(s->xchg) == AMT_XCHG_IPC;
        auto interior_beside = [&]() { return amt_slab_tile<T>(s, d->stream, in_lo, in_hi, true); };
        if (s->overlap) {
            AMT_HIP(hipEventRecord(s->inputs_final, d->stream));          // this sub-step's inputs are final
            AMT_HIP(hipStreamWaitEvent(s->comm_stream, s->inputs_final, 0));
            if (!exchange_first) {
                rc = interior_beside();                                    // interior overlaps the exchange
                if (rc) { join(); return rc; }
            }
        }
        if (s->skew_us > 0 && !amt_exchange_owns_skew(s->xchg)) {
            static const int wgs = [] { const char *e = getenv("AMT_SLAB_SKEW_WGS"); const int n = e ? atoi(e) : 1; return n > 1 ? n : 1; }();
            if (wgs > 1) {
                static const bool granted = hipFuncSetAttribute(reinterpret_cast<const void *>(amt_slab_delay_kernel),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
                hipLaunchKernelGGL(amt_slab_delay_kernel, dim3(wgs), dim3(256), granted ? 96 * 1024 : 48 * 1024, edge_stream,
                                   (unsigned long long)s->skew_us * 100ull);
            } else {
                hipLaunchKernelGGL(amt_slab_delay_kernel, dim3(1), dim3(1), 16, edge_stream, (unsigned long long)s->skew_us * 100ull);
            }
        }
        rc = amt_exchange_enqueue(s->xchg, edge_stream);
        if (rc) { join(); return rc; }
        if (s->overlap && exchange_first) {
            rc = interior_beside();
            if (rc) { join(); return rc; }
        }
        if (!s->overlap) {
            rc = amt_slab_tile<T>(s, d->stream, in_lo, in_hi);
            if (rc) return rc;
        }
        if (lo && hi && jhi > jlo && unclipped) {                           // both boundary rows in one launch
            AmtArgs<T> a;
            amt_domain_args<T>(d, a);
            a.jts = jlo; a.jte = jhi;
            rc = amt_device_call_edges<T>(edge_stream, d->variant, a);
            if (rc) { join(); return rc; }
        } else {                                                            // two one-row tiles, each clipped on its own
            if (lo) { rc = amt_slab_tile<T>(s, edge_stream, jlo, jlo < jhi ? jlo : jhi); if (rc) { join(); return rc; } }
            if (hi && (jhi > jlo || !lo)) { rc = amt_slab_tile<T>(s, edge_stream, jhi, jhi); if (rc) { join(); return rc; } }
        }
        // the sweep ends when the neighbours have this sweep's rows (RCCL: the sends of the group have completed; IPC: they
        // have pulled them) -- whatever the host model does to v, t_1, ... next cannot reach a neighbour's old read
        rc = amt_exchange_enqueue_release(s->xchg, edge_stream);
        if (rc) { join(); return rc; }
        if (s->overlap) {
            AMT_HIP(hipEventRecord(s->edges_done, s->comm_stream));
            AMT_HIP(hipStreamWaitEvent(d->stream, s->edges_done, 0));
        }
    }
    return AMT_OK;
}
}  // namespace

extern "C" int amt_slab_exchange(amt_slab *s)
{
    if (!s) return amt_fail(AMT_ERR_INVALID_ARG, "null slab");
    DeviceScope scope(s->dom->device);
    AMT_HIP(hipEventRecord(s->inputs_final, s->dom->stream));
    AMT_HIP(hipStreamWaitEvent(s->comm_stream, s->inputs_final, 0));
    int rc = amt_exchange_enqueue(s->xchg, s->comm_stream);
    if (rc == AMT_OK) rc = amt_exchange_enqueue_release(s->xchg, s->comm_stream);
    if (rc) return rc;
    AMT_HIP(hipEventRecord(s->edges_done, s->comm_stream));
    AMT_HIP(hipStreamWaitEvent(s->dom->stream, s->edges_done, 0));
    return AMT_OK;
}

extern "C" int amt_slab_step(amt_slab *s, int n_sweeps)
{
    if (!s || n_sweeps < 0) return amt_fail(AMT_ERR_INVALID_ARG, "bad step argument");
    DeviceScope scope(s->dom->device);
    return s->dom->dtype_bytes == 8 ? amt_slab_step_t<double>(s, n_sweeps) : amt_slab_step_t<float>(s, n_sweeps);
}

extern "C" int amt_slab_step_timed(amt_slab *s, int n_sweeps, float *ms_total)
{
    if (!s || n_sweeps < 0) return amt_fail(AMT_ERR_INVALID_ARG, "bad step argument");
    DeviceScope scope(s->dom->device);
    AMT_HIP(hipEventRecord(s->t0, s->dom->stream));
    int rc = amt_slab_step(s, n_sweeps);
    if (rc) return rc;
    AMT_HIP(hipEventRecord(s->t1, s->dom->stream));
    AMT_HIP(hipEventSynchronize(s->t1));
    float ms = 0.f;
    AMT_HIP(hipEventElapsedTime(&ms, s->t0, s->t1));
    if (ms_total) *ms_total = ms;
    return AMT_OK;
}

extern "C" int amt_slab_set_skew_us(amt_slab *s, int microseconds)
{
    if (!s || microseconds < 0) return amt_fail(AMT_ERR_INVALID_ARG, "bad skew argument");
    s->skew_us = microseconds;
    amt_exchange_set_skew_us(s->xchg, microseconds);      // the IPC transport carries it inside its waiting kernel
    return AMT_OK;
}

extern "C" int amt_slab_sync(amt_slab *s)
{
    if (!s) return amt_fail(AMT_ERR_INVALID_ARG, "null slab");
    DeviceScope scope(s->dom->device);
    AMT_HIP(hipStreamSynchronize(s->comm_stream));
    AMT_HIP(hipStreamSynchronize(s->dom->stream));
    return amt_exchange_check(s->xchg);          // a device-side wait that gave up (IPC) is reported here
}

extern "C" const char *amt_slab_transport(const amt_slab *s)
{
    if (!s || !amt_exchange_active(s->xchg)) return "none";
    return amt_exchange_transport(s->xchg) == AMT_XCHG_IPC ? "ipc" : "rccl";
}

extern "C" const char *amt_slab_pull_mode(const amt_slab *s) { return s ? amt_exchange_pull_mode(s->xchg) : ""; }

extern "C" long amt_slab_halo_bytes(const amt_slab *s)
{
    if (!s) return 0;
    size_t sent = 0, received = 0;
    amt_exchange_bytes(s->xchg, &sent, &received);
    return (long)(sent + received);
}


// What the transport itself says about this rank (ncclCommUserRank / ncclCommCount; the ranks attached to the IPC block): a
// bench line can then show that `world` ranks really joined.  Without a communicator: 0 of 1.
extern "C" int amt_slab_comm_info(const amt_slab *s, int *rank, int *world)
{
    if (!s) return amt_fail(AMT_ERR_INVALID_ARG, "null slab");
    return amt_exchange_info(s->xchg, rank, world);
}

// max over the ranks of *x (in place); also a barrier: every rank's streams are drained first and
// nobody returns before all have contributed.  For reporting only (the max-over-ranks sweep time
// of a host without MPI) -- the sweep itself uses no collective.
extern "C" int amt_slab_max(amt_slab *s, double *x)
{
    if (!s || !x) return amt_fail(AMT_ERR_INVALID_ARG, "bad reduction argument");
    DeviceScope scope(s->dom->device);
    AMT_HIP(hipStreamSynchronize(s->dom->stream));
    AMT_HIP(hipStreamSynchronize(s->comm_stream));
    return amt_exchange_max(s->xchg, x, s->comm_stream);
}

extern "C" int amt_slab_barrier(amt_slab *s)
{
    double zero = 0.0;
    return amt_slab_max(s, &zero);
}
