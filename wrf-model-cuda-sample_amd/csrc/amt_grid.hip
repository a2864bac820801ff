// amt_grid.hip -- advance_mu_t on patch (ri, rj) of a pi x pj decomposition in i AND j, one process per GPU: amt_grid_*
// (include/amt_advance_mu_t.h section 5b), and -- as its pi = 1 case -- the j-slab stepper amt_slab_*
// (section 5; the reference splits j over its GPUs inside one process with host-sourced halos,
// advance_mu_t_no_async.cu:108-162).
//
// What crosses a patch boundary (the stencil reads (i+-1, j) and (i, j+-1) only: no diagonal neighbours):
//   from above  (j+1): row jte+1 of v, v_1, t_1, muv, msfvx_inv      module_small_step_em.f90:143-144, :241
//   from below  (j-1): row jts-1 of t_1                               :242
//   from right  (i+1): column ite+1 of u, u_1, t_1, muu, msfuy        :145-146, :244
//   from left   (i-1): column its-1 of t_1                            :245
// A j row of the (i,k,j) layout is one contiguous run and travels in place.  A column is kdim*jdim elements at stride idim:
// one HIP kernel gathers the columns a patch sends into two contiguous buffers (one per direction), the buffers travel like
// rows, one kernel scatters what arrived into the halo columns.  The exchange engine of amt_comm.h carries both.
#include "amt_comm.h"
#include <vector>

struct amt_grid {
    amt_domain *dom = nullptr;
    int ri = 0, rj = 0, pi = 1, pj = 1, rank = 0, world = 1;
    int left = -1, right = -1, below = -1, above = -1;      // neighbour ranks, -1 = none
    bool overlap = true;
    AmtExchange *xchg = nullptr;
    hipStream_t comm_stream = nullptr;
    hipStream_t col_stream[2] = {nullptr, nullptr};         // the two boundary columns run beside the boundary rows
    hipEvent_t halos_in = nullptr, col_done[2] = {nullptr, nullptr};
    hipEvent_t inputs_final = nullptr, edges_done = nullptr, t0 = nullptr, t1 = nullptr;
    int skew_us = 0;                                        // test hook: the neighbours' rows arrive this late
    unsigned long long packs = 0;                           // exchanges begun (what AMT_TEST_FAULT skip_pack / skip_unpack count)
    // packed columns: what goes to the left / right neighbour, what came from the right / left one
    void *to_left = nullptr, *to_right = nullptr, *from_right = nullptr, *from_left = nullptr;
};
struct amt_slab {
    amt_grid g;
};

namespace {
const int kRowsFromAbove[] = {AMT_F_V, AMT_F_V_1, AMT_F_T_1, AMT_F_MUV, AMT_F_MSFVX_INV};
const int kRowsFromBelow[] = {AMT_F_T_1};
const int kColsFromRight[] = {AMT_F_U, AMT_F_U_1, AMT_F_T_1, AMT_F_MUU, AMT_F_MSFUY};       // 3-D ones first
const int kColsFromLeft[] = {AMT_F_T_1};

// One launch gathers (scatter = 0) or scatters (1) up to six columns: job q moves the `count` elements of memory column `col`
// of an array with rows of `idim` elements -- element e of the column is array[e * idim + col] -- from / to the contiguous run
// `packed`.  Lanes run along e (levels and rows): the packed side is coalesced, the array side touches one line per element,
// which is what a column is.
template <typename W>
struct AmtColumnJobs {
    W *array[6];
    W *packed[6];
    long col[6];
    long count[6];
    long idim;
    int n, scatter;
};
template <typename W>
__global__ __launch_bounds__(256) void amt_grid_columns(AmtColumnJobs<W> jobs)
{
    const int q = blockIdx.y;
    W *array = jobs.array[q] + jobs.col[q];
    W *packed = jobs.packed[q];
    const long n = jobs.count[q];
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        if (jobs.scatter) array[e * jobs.idim] = packed[e];
        else packed[e] = array[e * jobs.idim];
    }
}

struct ColumnPlan {
    size_t rows3, rows2, es;
    size_t bytes_from_right() const { return (3 * rows3 + 2 * rows2) * es; }      // u, u_1, t_1, muu, msfuy
    size_t bytes_from_left() const { return rows3 * es; }                          // t_1
};
ColumnPlan column_plan(const amt_domain *d)
{
    const size_t kdim = d->kme - d->kms + 1, jdim = d->jme - d->jms + 1;
    return ColumnPlan{kdim * jdim, jdim, (size_t)d->dtype_bytes};
}

template <typename W>
int grid_columns(amt_grid *g, hipStream_t stream, bool scatter)
{
    amt_domain *d = g->dom;
    const ColumnPlan cp = column_plan(d);
    AmtColumnJobs<W> jobs{};
    jobs.idim = d->ime - d->ims + 1;
    jobs.scatter = scatter ? 1 : 0;
    auto add = [&](int field, long col, void *packed, size_t offset_elems) {
        const int q = jobs.n++;
        jobs.array[q] = static_cast<W *>(d->field[field]);
        jobs.col[q] = col;
        jobs.count[q] = (long)(amt_field_rank(field) == 3 ? cp.rows3 : cp.rows2);
        jobs.packed[q] = static_cast<W *>(packed) + offset_elems;
    };
    const long c_first = d->its - d->ims, c_last = d->ite - d->ims;
    const int side_a = scatter ? g->right : g->left;      // gather: what the LEFT neighbour needs; scatter: what came from the RIGHT
    const int side_b = scatter ? g->left : g->right;
    if (side_a >= 0) {
        size_t off = 0;
        for (int f : kColsFromRight) {
            add(f, scatter ? c_last + 1 : c_first, scatter ? g->from_right : g->to_left, off);
            off += amt_field_rank(f) == 3 ? cp.rows3 : cp.rows2;
        }
    }
    if (side_b >= 0)
        for (int f : kColsFromLeft) add(f, scatter ? c_first - 1 : c_last, scatter ? g->from_left : g->to_right, 0);
    if (jobs.n == 0) return AMT_OK;
    long blocks = ((long)cp.rows3 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(amt_grid_columns<W>, dim3((unsigned)blocks, (unsigned)jobs.n), dim3(256), 0, stream, jobs);
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}
int grid_pack(amt_grid *g, hipStream_t s)
{
    if (amt_test_fault("skip_pack", ++g->packs)) return AMT_OK;
    return g->dom->dtype_bytes == 8 ? grid_columns<uint64_t>(g, s, false) : grid_columns<uint32_t>(g, s, false);
}
int grid_unpack(amt_grid *g, hipStream_t s)
{
    if (amt_test_fault("skip_unpack", g->packs)) return AMT_OK;
    return g->dom->dtype_bytes == 8 ? grid_columns<uint64_t>(g, s, true) : grid_columns<uint32_t>(g, s, true);
}

// Test hook (amt_*_set_skew_us, RCCL transport): holds the communication stream for `ticks` of the 100 MHz real-time counter, so
// that the exchange behind it starts -- and the neighbours' rows arrive -- that much late.  AMT_SLAB_SKEW_WGS=n (default 1) gives
// the delay the footprint of RCCL's waiting send/recv kernel: n workgroups of 256 threads, each with enough LDS to have a compute
// unit to itself.  (The IPC transport carries the delay inside its own waiting kernel.)
__global__ void amt_grid_delay_kernel(unsigned long long ticks)
{
    extern __shared__ unsigned char amt_delay_lds[];
    if (threadIdx.x == 0) amt_delay_lds[0] = 0;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

template <typename T>
int grid_tile(amt_grid *g, hipStream_t stream, int its, int ite, int jts, int jte, bool beside_the_exchange = false, bool thin_column = false)
{
    if (jte < jts || ite < its) return AMT_OK;
    AmtArgs<T> a;
    amt_domain_args<T>(g->dom, a);
    a.its = its; a.ite = ite; a.jts = jts; a.jte = jte;
    // A boundary COLUMN is one column wide: the march kernel streams a whole tile for it; the column kernel -- one lane per
    // column, the same bits -- would touch one line per element instead.
    // Measured: the column kernel is the SLOWER one here (2048^2 patch: +18 % against +8.6 % per sweep, one active lane per
    // workgroup walking its levels in sequence): off unless AMT_GRID_THIN_COLUMNS=1.
    static const bool thin_env = [] { const char *e = getenv("AMT_GRID_THIN_COLUMNS"); return e && *e && atoi(e) != 0; }();
    if (thin_column && thin_env && g->dom->variant == AMT_VARIANT_AUTO) return amt_device_call<T>(stream, AMT_VARIANT_COLUMN, a);
    // A launch that is ONE round of workgroups (the launcher's choice for a patch on its own) holds every compute unit
    // until it ends; beside the exchange the interior is planned as amt_march_set_beside says (profiles/r05_slab_ab.md).
    return beside_the_exchange ? amt_device_call_shared<T>(stream, g->dom->variant, a) : amt_device_call<T>(stream, g->dom->variant, a);
}

// the cells that read a neighbour's data, after the halos are in: boundary rows over the patch's whole width (they own the
// corners), boundary columns over the rows in between; every tile is clipped on its own by the routine's window rule
template <typename T>
int grid_edges(amt_grid *g, hipStream_t edge_stream, bool lo, bool hi, bool lf, bool rt, bool unclipped, int in_jlo, int in_jhi)
{
    amt_domain *d = g->dom;
    const int ilo = d->its, ihi = d->ite, jlo = d->jts, jhi = d->jte;
    int rc = AMT_OK;
    // The boundary columns are independent of the rows and of each other: they go to streams of their own (three small
    // launches, the chip has room for all of them at once) behind the event "halos are in", and join the edge stream.
    const bool fork = (lf || rt) && g->col_stream[0] && edge_stream == g->comm_stream;
    if (fork) AMT_HIP(hipEventRecord(g->halos_in, edge_stream));
    int used = 0;
    auto column = [&](int c) -> int {
        hipStream_t st = fork ? g->col_stream[used] : edge_stream;
        if (fork) AMT_HIP(hipStreamWaitEvent(st, g->halos_in, 0));
        const int rc2 = grid_tile<T>(g, st, c, c, in_jlo, in_jhi, false, true);
        if (fork) (void)hipEventRecord(g->col_done[used++], st);
        return rc2;
    };
    if (lf) rc = column(ilo < ihi ? ilo : ihi);
    if (rc == AMT_OK && rt && (ihi > ilo || !lf)) rc = column(ihi);
    if (rc == AMT_OK) {
        if (lo && hi && jhi > jlo && unclipped) {                       // both rows in one launch
            AmtArgs<T> a;
            amt_domain_args<T>(d, a);
            a.jts = jlo; a.jte = jhi;
            rc = amt_device_call_edges<T>(edge_stream, d->variant, a);
        } else {                                                        // one-row tiles
            if (lo) rc = grid_tile<T>(g, edge_stream, ilo, ihi, jlo, jlo < jhi ? jlo : jhi);
            if (rc == AMT_OK && hi && (jhi > jlo || !lo)) rc = grid_tile<T>(g, edge_stream, ilo, ihi, jhi, jhi);
        }
    }
    for (int q = 0; q < used; ++q) (void)hipStreamWaitEvent(edge_stream, g->col_done[q], 0);      // always joined, also after an error
    return rc;
}

template <typename T>
int grid_step_t(amt_grid *g, int n_sweeps)
{
    amt_domain *d = g->dom;
    const int ilo = d->its, ihi = d->ite, jlo = d->jts, jhi = d->jte;
    const bool lo = g->below >= 0, hi = g->above >= 0, lf = g->left >= 0, rt = g->right >= 0;
    // the rows the routine really updates in this patch (module_small_step_em.f90:91-106): with specified / nested boundaries
    // the first / last row of an outermost patch is clipped away.  Every tile below is clipped on its own by the same rule;
    // only the one-launch path for both boundary rows computes rows j_start and j_end of the CLIPPED window, so it is taken only
    // when nothing is clipped (always, for a patch with a neighbour on that side -- except in loopback, where the rank is its
    // own neighbour on a patch that touches the domain edge).
    const AmtWindow wclip = amt_window(d->periodic_x, d->specified, d->nested, d->ids, d->ide, d->jds, d->jde,
                                       ilo, ihi, jlo, jhi, d->kts, d->kte);
    const bool unclipped = wclip.j_start == jlo && wclip.j_end == jhi;
    // a failure between the fork (inputs_final) and the join (edges_done) must not leave the streams apart
    auto join = [&]() {
        (void)hipEventRecord(g->edges_done, g->comm_stream);
        (void)hipStreamWaitEvent(d->stream, g->edges_done, 0);
    };
    const bool none = !lo && !hi && !lf && !rt;
    // cells that read a neighbour's data: rows jlo / jhi, columns ilo / ihi; the rest is interior
    const int in_jlo = jlo + (lo ? 1 : 0), in_jhi = jhi - (hi ? 1 : 0);
    const int in_ilo = ilo + (lf ? 1 : 0), in_ihi = ihi - (rt ? 1 : 0);
    const bool ipc = amt_exchange_transport(g->xchg) == AMT_XCHG_IPC && amt_exchange_active(g->xchg);
    static const bool host_wait_env = [] { const char *e = getenv("AMT_IPC_HOST_WAIT"); return !(e && *e && atoi(e) == 0); }();
    static const int order_env = [] { const char *e = getenv("AMT_SLAB_EXCHANGE_FIRST"); return e && *e ? atoi(e) : -1; }();
    auto delay_for_the_test_hook = [&](hipStream_t stream) {
        if (g->skew_us <= 0 || amt_exchange_owns_skew(g->xchg)) return;
        static const int wgs = [] { const char *e = getenv("AMT_SLAB_SKEW_WGS"); const int n = e ? atoi(e) : 1; return n > 1 ? n : 1; }();
        if (wgs > 1) {
            static const bool granted = hipFuncSetAttribute(reinterpret_cast<const void *>(amt_grid_delay_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
            hipLaunchKernelGGL(amt_grid_delay_kernel, dim3(wgs), dim3(256), granted ? 96 * 1024 : 48 * 1024, stream, (unsigned long long)g->skew_us * 100ull);
        } else {
            hipLaunchKernelGGL(amt_grid_delay_kernel, dim3(1), dim3(1), 16, stream, (unsigned long long)g->skew_us * 100ull);
        }
    };
    for (int sweep = 0; sweep < n_sweeps; ++sweep) {
        // a device-side wait of an EARLIER sweep that gave up (IPC transport: a neighbour that never posted or never pulled):
        // that sweep's halo rows were not valid, so nothing is built on top of it -- the step fails here, not only in *_sync
        int rc = amt_exchange_check(g->xchg);
        if (rc) return rc;
        if (none) {                                                        // a world of one: the plain launch
            rc = grid_tile<T>(g, d->stream, ilo, ihi, jlo, jhi);
            if (rc) return rc;
            continue;
        }
        if (!g->overlap) {
            // (1) NO OVERLAP: the halos first, then the WHOLE patch as one launch (nothing to split when nothing runs beside)
            rc = grid_pack(g, d->stream);
            if (rc == AMT_OK) delay_for_the_test_hook(d->stream);
            if (rc == AMT_OK) rc = amt_exchange_enqueue(g->xchg, d->stream, true);
            if (rc == AMT_OK) rc = grid_unpack(g, d->stream);
            if (rc == AMT_OK) rc = grid_tile<T>(g, d->stream, ilo, ihi, jlo, jhi);
            if (rc == AMT_OK) rc = amt_exchange_enqueue_release(g->xchg, d->stream);
            if (rc) return rc;
            continue;
        }
        if (ipc && host_wait_env) {
            // (2) HOST-WAITED (IPC transport, default; AMT_IPC_HOST_WAIT=0 for (3)).  A march workgroup takes a compute unit whole,
            // so nothing can run BESIDE the interior (profiles/r05_slab_ab.md): here nothing of the exchange holds a unit while
            // the interior runs, the interior is planned on its own -- one round where it can be one, full efficiency -- and the
            // neighbours' lateness hides behind ALL of it:
            //   domain stream: [gather columns] -> refresh the staging copies -> post "rows final n" (one wave) -> interior
            //   host:          poll the mailbox until every neighbour has posted n (the call returns after that: per sub-step,
            //                  as a host that exchanges by MPI would wait)
            //   comm stream:   pull (copy engine between GPUs; one kernel on a shared device) -> post "pulled n" -> [scatter
            //                  columns] -> boundary rows / columns -> wait until the neighbours have pulled -> join
            // The boundary tiles get their compute units when the interior's workgroups end: they run right behind it.
            rc = grid_pack(g, d->stream);
            if (rc == AMT_OK) rc = amt_exchange_enqueue_post(g->xchg, d->stream);
            if (rc) return rc;
            if (hipEventRecord(g->inputs_final, d->stream) != hipSuccess || hipStreamWaitEvent(g->comm_stream, g->inputs_final, 0) != hipSuccess)
                return amt_fail(AMT_ERR_HIP, "amt_grid_step: cannot fork the communication stream: %s", hipGetErrorString(hipGetLastError()));
            rc = grid_tile<T>(g, d->stream, in_ilo, in_ihi, in_jlo, in_jhi);       // on its own: the launcher's best plan
            if (rc == AMT_OK) rc = amt_exchange_host_wait(g->xchg);
            if (rc == AMT_OK) rc = amt_exchange_enqueue_pull(g->xchg, g->comm_stream);
            if (rc == AMT_OK) rc = grid_unpack(g, g->comm_stream);
            if (rc == AMT_OK) rc = grid_edges<T>(g, g->comm_stream, lo, hi, lf, rt, unclipped, in_jlo, in_jhi);
            if (rc == AMT_OK) rc = amt_exchange_enqueue_release(g->xchg, g->comm_stream);
            join();                                                        // also after an error: the streams never stay apart
            if (rc) return rc;
            continue;
        }
        // (3) DEVICE-WAITED (RCCL; IPC with AMT_IPC_HOST_WAIT=0): the exchange on the communication stream, the interior BESIDE it
        // on the domain's stream, planned by amt_march_set_beside (rounds, reserved units).  Whatever the communication stream
        // launches once the interior is out starts only where an interior workgroup ends; the IPC transport's waiting kernel
        // therefore goes out FIRST (it has its units from the start of the sweep), RCCL's send/recv kernel -- which holds its
        // units for as long as it waits -- after the interior (AMT_SLAB_EXCHANGE_FIRST=0|1 overrides either).
        const bool exchange_first = order_env >= 0 ? order_env != 0 : ipc;
        if (hipEventRecord(g->inputs_final, d->stream) != hipSuccess || hipStreamWaitEvent(g->comm_stream, g->inputs_final, 0) != hipSuccess)
            return amt_fail(AMT_ERR_HIP, "amt_grid_step: cannot fork the communication stream: %s", hipGetErrorString(hipGetLastError()));
        if (!exchange_first) rc = grid_tile<T>(g, d->stream, in_ilo, in_ihi, in_jlo, in_jhi, true);
        if (rc == AMT_OK) delay_for_the_test_hook(g->comm_stream);
        if (rc == AMT_OK) rc = grid_pack(g, g->comm_stream);               // the columns this patch sends, gathered
        if (rc == AMT_OK) rc = amt_exchange_enqueue(g->xchg, g->comm_stream);
        if (rc == AMT_OK && exchange_first) rc = grid_tile<T>(g, d->stream, in_ilo, in_ihi, in_jlo, in_jhi, true);
        if (rc == AMT_OK) rc = grid_unpack(g, g->comm_stream);             // the columns that arrived, scattered into the halo
        if (rc == AMT_OK) rc = grid_edges<T>(g, g->comm_stream, lo, hi, lf, rt, unclipped, in_jlo, in_jhi);
        // the sweep ends when the neighbours have this sweep's rows (RCCL: the sends of the group have completed; IPC: they
        // have pulled them)
        if (rc == AMT_OK) rc = amt_exchange_enqueue_release(g->xchg, g->comm_stream);
        join();
        if (rc) return rc;
    }
    return AMT_OK;
}

void grid_teardown(amt_grid *g)
{
    DeviceScope scope(g->dom ? g->dom->device : 0);
    if (g->comm_stream) (void)hipStreamSynchronize(g->comm_stream);
    (void)amt_exchange_destroy(g->xchg);
    g->xchg = nullptr;
    for (hipStream_t st : {g->col_stream[0], g->col_stream[1]})
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : {g->inputs_final, g->edges_done, g->t0, g->t1, g->halos_in, g->col_done[0], g->col_done[1]})
        if (e) (void)hipEventDestroy(e);
    if (g->comm_stream) (void)hipStreamDestroy(g->comm_stream);
    for (void *q : {g->to_left, g->to_right, g->from_right, g->from_left})
        if (q) (void)hipFree(q);
}

// loop_i / loop_j: one-rank test mode -- the rank is its own neighbour across i / across j
int grid_setup(amt_grid *g, amt_domain *dom, int ri, int rj, int pi, int pj, const void *unique_id, int flags, bool loop_i, bool loop_j)
{
    if (!dom || pi < 1 || pj < 1 || ri < 0 || ri >= pi || rj < 0 || rj >= pj) return amt_fail(AMT_ERR_INVALID_ARG, "bad patch index");
    const int world = pi * pj, rank = rj * pi + ri;
    if ((loop_i || loop_j) && world != 1) return amt_fail(AMT_ERR_INVALID_ARG, "loopback is a one-rank test mode");
    const bool comm_needed = world > 1 || loop_i || loop_j;
    if (comm_needed && !unique_id) return amt_fail(AMT_ERR_INVALID_ARG, "a communicator needs the unique id");
    g->dom = dom; g->ri = ri; g->rj = rj; g->pi = pi; g->pj = pj; g->rank = rank; g->world = world;
    g->overlap = !(flags & AMT_SLAB_NO_OVERLAP);
    g->left = loop_i ? rank : ri > 0 ? rank - 1 : -1;
    g->right = loop_i ? rank : ri < pi - 1 ? rank + 1 : -1;
    g->below = loop_j ? rank : rj > 0 ? rank - pi : -1;
    g->above = loop_j ? rank : rj < pj - 1 ? rank + pi : -1;
    if ((g->below >= 0 || g->above >= 0) && (dom->jts - 1 < dom->jms || dom->jte + 1 > dom->jme))
        return amt_fail(AMT_ERR_PRECONDITION, "a patch holds one halo row below jts and above jte");
    if ((g->left >= 0 || g->right >= 0) && (dom->its - 1 < dom->ims || dom->ite + 1 > dom->ime))
        return amt_fail(AMT_ERR_PRECONDITION, "a patch holds one halo column left of its and right of ite");
    int transport = (flags & AMT_SLAB_TRANSPORT_IPC) ? AMT_XCHG_IPC : AMT_XCHG_RCCL;
    if (const char *e = getenv("AMT_SLAB_TRANSPORT")) {                  // hosts that cannot pass the flag (the Fortran drivers)
        if (!strcmp(e, "ipc")) transport = AMT_XCHG_IPC;
        else if (!strcmp(e, "rccl")) transport = AMT_XCHG_RCCL;
        else if (*e) return amt_fail(AMT_ERR_INVALID_ARG, "AMT_SLAB_TRANSPORT must be rccl or ipc, not '%s'", e);
    }
    DeviceScope scope(dom->device);
    // the communication stream outranks the domain's: where a compute unit is free, the exchange and the edge tiles get it
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    hipError_t e = hipStreamCreateWithPriority(&g->comm_stream, hipStreamNonBlocking, prio_high);
    for (hipEvent_t *ev : {&g->inputs_final, &g->edges_done})
        if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    for (hipEvent_t *ev : {&g->t0, &g->t1})
        if (e == hipSuccess) e = hipEventCreate(ev);
    if (g->left >= 0 || g->right >= 0) {
        for (hipStream_t *st : {&g->col_stream[0], &g->col_stream[1]})
            if (e == hipSuccess) e = hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_high);
        for (hipEvent_t *ev : {&g->halos_in, &g->col_done[0], &g->col_done[1]})
            if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    }
    const ColumnPlan cp = column_plan(dom);
    if (e == hipSuccess && g->left >= 0) e = hipMalloc(&g->to_left, cp.bytes_from_right());
    if (e == hipSuccess && g->right >= 0) e = hipMalloc(&g->from_right, cp.bytes_from_right());
    if (e == hipSuccess && g->right >= 0) e = hipMalloc(&g->to_right, cp.bytes_from_left());
    if (e == hipSuccess && g->left >= 0) e = hipMalloc(&g->from_left, cp.bytes_from_left());
    if (e != hipSuccess) return amt_fail(e == hipErrorOutOfMemory ? AMT_ERR_ALLOC : AMT_ERR_HIP, "amt_grid_create: %s", hipGetErrorString(e));
    // the segments of one exchange.  Per pair of ranks the order of the sends is the order of the receives on the other side.
    const size_t idim = dom->ime - dom->ims + 1, kdim = dom->kme - dom->kms + 1;
    auto row = [&](int f, int j, int peer) {
        const size_t count = amt_field_rank(f) == 3 ? idim * kdim : idim;
        return AmtSeg{static_cast<char *>(dom->field[f]) + (size_t)(j - dom->jms) * count * dom->dtype_bytes, count * dom->dtype_bytes, peer};
    };
    std::vector<AmtSeg> sends, recvs;
    if (g->below >= 0) for (int f : kRowsFromAbove) sends.push_back(row(f, dom->jts, g->below));
    if (g->above >= 0) for (int f : kRowsFromBelow) sends.push_back(row(f, dom->jte, g->above));
    if (g->left >= 0) sends.push_back(AmtSeg{g->to_left, cp.bytes_from_right(), g->left});
    if (g->right >= 0) sends.push_back(AmtSeg{g->to_right, cp.bytes_from_left(), g->right});
    if (g->above >= 0) for (int f : kRowsFromAbove) recvs.push_back(row(f, dom->jte + 1, g->above));
    if (g->below >= 0) for (int f : kRowsFromBelow) recvs.push_back(row(f, dom->jts - 1, g->below));
    if (g->right >= 0) recvs.push_back(AmtSeg{g->from_right, cp.bytes_from_right(), g->right});
    if (g->left >= 0) recvs.push_back(AmtSeg{g->from_left, cp.bytes_from_left(), g->left});
    return amt_exchange_create(&g->xchg, transport, rank, world, unique_id, dom->device, sends.data(), (int)sends.size(),
                               recvs.data(), (int)recvs.size(), loop_i || loop_j);
}

int grid_exchange_only(amt_grid *g)
{
    DeviceScope scope(g->dom->device);
    AMT_HIP(hipEventRecord(g->inputs_final, g->dom->stream));
    AMT_HIP(hipStreamWaitEvent(g->comm_stream, g->inputs_final, 0));
    int rc = grid_pack(g, g->comm_stream);
    if (rc == AMT_OK) rc = amt_exchange_enqueue(g->xchg, g->comm_stream, true);
    if (rc == AMT_OK) rc = grid_unpack(g, g->comm_stream);
    if (rc == AMT_OK) rc = amt_exchange_enqueue_release(g->xchg, g->comm_stream);
    if (rc) return rc;
    AMT_HIP(hipEventRecord(g->edges_done, g->comm_stream));
    AMT_HIP(hipStreamWaitEvent(g->dom->stream, g->edges_done, 0));
    return AMT_OK;
}

int grid_step(amt_grid *g, int n_sweeps)
{
    DeviceScope scope(g->dom->device);
    return g->dom->dtype_bytes == 8 ? grid_step_t<double>(g, n_sweeps) : grid_step_t<float>(g, n_sweeps);
}

int grid_step_timed(amt_grid *g, int n_sweeps, float *ms_total)
{
    DeviceScope scope(g->dom->device);
    AMT_HIP(hipEventRecord(g->t0, g->dom->stream));
    int rc = grid_step(g, n_sweeps);
    if (rc) return rc;
    AMT_HIP(hipEventRecord(g->t1, g->dom->stream));
    AMT_HIP(hipEventSynchronize(g->t1));
    float ms = 0.f;
    AMT_HIP(hipEventElapsedTime(&ms, g->t0, g->t1));
    if (ms_total) *ms_total = ms;
    return amt_exchange_check(g->xchg);          // the timed sweeps are complete: report a wait that gave up inside them
}

int grid_sync(amt_grid *g)
{
    DeviceScope scope(g->dom->device);
    AMT_HIP(hipStreamSynchronize(g->comm_stream));
    AMT_HIP(hipStreamSynchronize(g->dom->stream));
    return amt_exchange_check(g->xchg);          // a device-side wait that gave up (IPC) is reported here
}

long grid_halo_bytes(const amt_grid *g)
{
    size_t sent = 0, received = 0;
    amt_exchange_bytes(g->xchg, &sent, &received);
    return (long)(sent + received);
}

const char *grid_transport(const amt_grid *g)
{
    if (!amt_exchange_active(g->xchg)) return "none";
    return amt_exchange_transport(g->xchg) == AMT_XCHG_IPC ? "ipc" : "rccl";
}

// max over the ranks of *x (in place); also a barrier: every rank's streams are drained first and nobody returns before all
// have contributed.  For reporting only (the max-over-ranks sweep time of a host without MPI) -- the sweep itself uses no
// collective.
int grid_max(amt_grid *g, double *x)
{
    DeviceScope scope(g->dom->device);
    AMT_HIP(hipStreamSynchronize(g->dom->stream));
    AMT_HIP(hipStreamSynchronize(g->comm_stream));
    return amt_exchange_max(g->xchg, x, g->comm_stream);
}

template <typename H>
int create_handle(H **out, amt_domain *dom, int ri, int rj, int pi, int pj, const void *unique_id, int flags, bool loop_i, bool loop_j, amt_grid *(*grid_of)(H *))
{
    if (!out) return amt_fail(AMT_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    H *h = new (std::nothrow) H;
    if (!h) return amt_fail(AMT_ERR_ALLOC, "host allocation failed");
    const int rc = grid_setup(grid_of(h), dom, ri, rj, pi, pj, unique_id, flags, loop_i, loop_j);
    if (rc) {
        const std::string keep = amt_last_error();       // the teardown must not lose the diagnosis
        if (grid_of(h)->dom) grid_teardown(grid_of(h));
        delete h;
        return amt_fail(rc, "%s", keep.c_str());
    }
    *out = h;
    return AMT_OK;
}
amt_grid *self(amt_grid *g) { return g; }
amt_grid *inner(amt_slab *s) { return &s->g; }
}  // namespace

// ---------------------------------------------------------------------------
// amt_grid_*: patch (ri, rj) of pi x pj; rank = rj * pi + ri
// ---------------------------------------------------------------------------
extern "C" int amt_grid_create(amt_grid **out, amt_domain *dom, int ri, int rj, int pi, int pj, const void *unique_id, int flags)
{
    const bool loop = (flags & AMT_SLAB_LOOPBACK) != 0;
    return create_handle<amt_grid>(out, dom, ri, rj, pi, pj, unique_id, flags, loop, loop, self);
}
extern "C" int amt_grid_destroy(amt_grid *g)
{
    if (!g) return AMT_OK;
    grid_teardown(g);
    delete g;
    return AMT_OK;
}
extern "C" int amt_grid_exchange(amt_grid *g) { return g ? grid_exchange_only(g) : amt_fail(AMT_ERR_INVALID_ARG, "null grid"); }
extern "C" int amt_grid_step(amt_grid *g, int n) { return g && n >= 0 ? grid_step(g, n) : amt_fail(AMT_ERR_INVALID_ARG, "bad step argument"); }
extern "C" int amt_grid_step_timed(amt_grid *g, int n, float *ms) { return g && n >= 0 ? grid_step_timed(g, n, ms) : amt_fail(AMT_ERR_INVALID_ARG, "bad step argument"); }
extern "C" int amt_grid_sync(amt_grid *g) { return g ? grid_sync(g) : amt_fail(AMT_ERR_INVALID_ARG, "null grid"); }
extern "C" long amt_grid_halo_bytes(const amt_grid *g) { return g ? grid_halo_bytes(g) : 0; }
extern "C" const char *amt_grid_transport(const amt_grid *g) { return g ? grid_transport(g) : "none"; }
extern "C" int amt_grid_comm_info(const amt_grid *g, int *rank, int *world) { return g ? amt_exchange_info(g->xchg, rank, world) : amt_fail(AMT_ERR_INVALID_ARG, "null grid"); }
extern "C" int amt_grid_max(amt_grid *g, double *x) { return g && x ? grid_max(g, x) : amt_fail(AMT_ERR_INVALID_ARG, "bad reduction argument"); }
extern "C" int amt_grid_barrier(amt_grid *g) { double zero = 0.0; return amt_grid_max(g, &zero); }
extern "C" int amt_grid_set_skew_us(amt_grid *g, int microseconds)
{
    if (!g || microseconds < 0) return amt_fail(AMT_ERR_INVALID_ARG, "bad skew argument");
    g->skew_us = microseconds;
    amt_exchange_set_skew_us(g->xchg, microseconds);      // the IPC transport carries it inside its waiting kernel
    return AMT_OK;
}

// ---------------------------------------------------------------------------
// amt_slab_*: the pi = 1 case (rank = rj, world = pj); loopback loops j only
// ---------------------------------------------------------------------------
extern "C" int amt_slab_create(amt_slab **out, amt_domain *dom, int rank, int world, const void *unique_id, int flags)
{
    if (world < 1 || rank < 0 || rank >= world) return amt_fail(AMT_ERR_INVALID_ARG, "bad slab argument");
    const bool loop = (flags & AMT_SLAB_LOOPBACK) != 0;
    if (loop && world != 1) return amt_fail(AMT_ERR_INVALID_ARG, "AMT_SLAB_LOOPBACK is a one-rank test mode");
    return create_handle<amt_slab>(out, dom, 0, rank, 1, world, unique_id, flags, false, loop, inner);
}
extern "C" int amt_slab_destroy(amt_slab *s)
{
    if (!s) return AMT_OK;
    grid_teardown(&s->g);
    delete s;
    return AMT_OK;
}
extern "C" int amt_slab_exchange(amt_slab *s) { return s ? grid_exchange_only(&s->g) : amt_fail(AMT_ERR_INVALID_ARG, "null slab"); }
extern "C" int amt_slab_step(amt_slab *s, int n) { return s && n >= 0 ? grid_step(&s->g, n) : amt_fail(AMT_ERR_INVALID_ARG, "bad step argument"); }
extern "C" int amt_slab_step_timed(amt_slab *s, int n, float *ms) { return s && n >= 0 ? grid_step_timed(&s->g, n, ms) : amt_fail(AMT_ERR_INVALID_ARG, "bad step argument"); }
extern "C" int amt_slab_sync(amt_slab *s) { return s ? grid_sync(&s->g) : amt_fail(AMT_ERR_INVALID_ARG, "null slab"); }
extern "C" long amt_slab_halo_bytes(const amt_slab *s) { return s ? grid_halo_bytes(&s->g) : 0; }
extern "C" const char *amt_slab_transport(const amt_slab *s) { return s ? grid_transport(&s->g) : "none"; }
extern "C" const char *amt_slab_pull_mode(const amt_slab *s) { return s ? amt_exchange_pull_mode(s->g.xchg) : ""; }
extern "C" const char *amt_grid_pull_mode(const amt_grid *g) { return g ? amt_exchange_pull_mode(g->xchg) : ""; }
extern "C" int amt_slab_comm_info(const amt_slab *s, int *rank, int *world) { return s ? amt_exchange_info(s->g.xchg, rank, world) : amt_fail(AMT_ERR_INVALID_ARG, "null slab"); }
extern "C" int amt_slab_max(amt_slab *s, double *x) { return s && x ? grid_max(&s->g, x) : amt_fail(AMT_ERR_INVALID_ARG, "bad reduction argument"); }
extern "C" int amt_slab_barrier(amt_slab *s) { double zero = 0.0; return amt_slab_max(s, &zero); }
extern "C" int amt_slab_set_skew_us(amt_slab *s, int microseconds)
{
    if (!s) return amt_fail(AMT_ERR_INVALID_ARG, "bad skew argument");
    return amt_grid_set_skew_us(&s->g, microseconds);
}
