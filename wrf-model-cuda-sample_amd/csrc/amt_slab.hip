// amt_slab.hip -- j-slab stepping with RCCL halos, amt_slab_* (include/amt_advance_mu_t.h section 5).
#include "amt_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <mutex>
#include <thread>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

// ---------------------------------------------------------------------------
// the native twin of patch.SlabStepper, for C / Fortran
//     hosts that run one process per GPU (the reference splits j over its
//     GPUs inside one process with host-sourced halos, advance_mu_t_no_async.cu:108-162).
//     RCCL is opened with dlopen on first use: the library has no link-time dependency on it
//     and single-GPU users never load it.
// ---------------------------------------------------------------------------
namespace {
struct AmtRccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
AmtRccl g_rccl;
std::mutex g_rccl_mutex;

int amt_rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.lib) return AMT_OK;
    const char *names[] = {getenv("AMT_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names)
        if (n && *n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return amt_fail(AMT_ERR_COMM, "cannot open librccl: %s", dlerror());
    AmtRccl r;
    r.lib = lib;
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(lib, name); ok = ok && p; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(lib); return amt_fail(AMT_ERR_COMM, "librccl lacks a send/recv entry point"); }
    g_rccl = r;
    return AMT_OK;
}
}  // namespace

#define AMT_NCCL(call)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return amt_fail(AMT_ERR_COMM, "%s failed: %s (%s:%d)", #call,                       \
                            g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", __FILE__, __LINE__); \
    } while (0)

static_assert(AMT_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "amt_comm_unique_id hands out an ncclUniqueId");

extern "C" int amt_set_device(int device)
{
    AMT_HIP(hipSetDevice(device));
    return AMT_OK;
}

extern "C" int amt_comm_unique_id(void *id_out)
{
    if (!id_out) return amt_fail(AMT_ERR_INVALID_ARG, "null id buffer");
    int rc = amt_rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    AMT_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return AMT_OK;
}

// Rendezvous for hosts without MPI: rank 0 creates the id and publishes it as `path` (written
// under a temporary name, then renamed) behind a header that carries the LAUNCH NONCE; the other
// ranks wait for a file whose nonce is theirs, read the id and acknowledge with `path.ack.<rank>`;
// rank 0 waits for the world-1 acknowledgements and removes every file.  A file left behind by an
// earlier launch (a crashed one: a completed one leaves nothing) carries another nonce and is
// never taken for this launch's, however recently it was written.
namespace {
struct AmtRendezvousHeader {
    char magic[8];
    uint64_t nonce;
};
const char kRvMagic[8] = {'A', 'M', 'T', 'U', 'I', 'D', '0', '2'};

uint64_t amt_fnv1a(uint64_t h, const void *data, size_t n)
{
    const unsigned char *q = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; ++i) { h ^= q[i]; h *= 1099511628211ull; }
    return h;
}
}  // namespace

// A value every process of ONE launch computes identically and two launches do not share:
// AMT_RENDEZVOUS_NONCE if set; else a job id the scheduler gives every rank of the job (SLURM_JOB_ID +
// step, PMI / PMIx / Open MPI job ids, LSB_JOBID, PBS_JOBID) -- such ranks need not share a parent
// process (one srun / orted daemon per node) -- together with MASTER_PORT; else the parent process (pid and
// start time from /proc: the ranks of a local launch are children of one launcher) together with MASTER_PORT.
// A LOCAL launcher inside either (TORCHELASTIC_RUN_ID is set: torchrun started the ranks of this node) adds
// its run id, its restart count and itself as the parent process: two torchrun launches inside one
// allocation, or an elastic restart on the same port, then differ although the scheduler's ids do not
// (without that, ranks >= 1 of the second launch accepted the file a crashed first one left).
// Ranks started by hand (a shell or ssh per rank) have neither a job id nor a common parent, and ranks of
// several torchrun agents that share the rendezvous file over a network file system have different parents:
// both must be given AMT_RENDEZVOUS_NONCE.  Never 0.
extern "C" uint64_t amt_comm_launch_nonce(void)
{
    uint64_t h = 1469598103934665603ull;
    static const char *const job_ids[] = {"SLURM_JOB_ID", "SLURM_STEP_ID", "PMI_JOBID", "PMI_ID_JOB", "PMIX_NAMESPACE",
                                          "OMPI_MCA_ess_base_jobid", "LSB_JOBID", "PBS_JOBID"};
    auto mix_env = [&](const char *name) {
        if (const char *t = getenv(name); t && *t) { h = amt_fnv1a(h, name, strlen(name)); h = amt_fnv1a(h, t, strlen(t)); }
    };
    auto mix_parent = [&] {
        const long ppid = (long)getppid();
        h = amt_fnv1a(h, &ppid, sizeof ppid);
        char statpath[64];
        snprintf(statpath, sizeof statpath, "/proc/%ld/stat", ppid);
        if (FILE *f = fopen(statpath, "r")) {
            char buf[1024];
            const size_t n = fread(buf, 1, sizeof buf - 1, f);
            fclose(f);
            buf[n] = 0;
            // field 22 (starttime) counted after the last ')' of the command name
            if (const char *q = strrchr(buf, ')')) {
                int field = 2;
                for (++q; *q && field < 22; ++q)
                    if (*q == ' ') ++field;
                const char *e = q;
                while (*e && *e != ' ') ++e;
                h = amt_fnv1a(h, q, (size_t)(e - q));
            }
        }
    };
    bool have_job = false;
    for (const char *name : job_ids)
        if (const char *t = getenv(name); t && *t) have_job = true;
    const char *elastic = getenv("TORCHELASTIC_RUN_ID");
    const bool local_launcher = elastic && *elastic;
    if (const char *s = getenv("AMT_RENDEZVOUS_NONCE"); s && *s) {
        h = amt_fnv1a(h, s, strlen(s));
        return h ? h : 1;
    }
    if (have_job)
        for (const char *name : job_ids) mix_env(name);
    if (local_launcher) {
        mix_env("TORCHELASTIC_RUN_ID");
        mix_env("TORCHELASTIC_RESTART_COUNT");
    }
    // the launcher's run id alone may be a fixed word ("none" for a static rendezvous): the parent process as well,
    // unless the scheduler's ids are all there is to agree on (its ranks have one daemon per node as parents)
    if (!have_job || local_launcher) mix_parent();
    mix_env("MASTER_PORT");
    return h ? h : 1;
}

extern "C" int amt_comm_rendezvous_file(const char *path, uint64_t nonce, int rank, int world,
                                        double timeout_s, void *id_out)
{
    if (!path || !*path || !id_out || rank < 0 || world < 1 || rank >= world)
        return amt_fail(AMT_ERR_INVALID_ARG, "bad rendezvous argument");
    if (nonce == 0) nonce = amt_comm_launch_nonce();
    const auto t0 = std::chrono::steady_clock::now();
    auto waited = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    auto ack_name = [&](int r) { return std::string(path) + ".ack." + std::to_string(r); };
    if (rank == 0) {
        (void)unlink(path);
        for (int r = 1; r < world; ++r) (void)unlink(ack_name(r).c_str());
        int rc = amt_comm_unique_id(id_out);
        if (rc) return rc;
        AmtRendezvousHeader hd;
        memcpy(hd.magic, kRvMagic, 8);
        hd.nonce = nonce;
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return amt_fail(AMT_ERR_COMM, "cannot write %s", tmp.c_str());
        const size_t n = fwrite(&hd, 1, sizeof hd, f) + fwrite(id_out, 1, AMT_UNIQUE_ID_BYTES, f);
        fclose(f);
        if (n != sizeof hd + AMT_UNIQUE_ID_BYTES || rename(tmp.c_str(), path) != 0)
            return amt_fail(AMT_ERR_COMM, "cannot publish %s", path);
        // wait until every other rank of THIS launch has the id, then leave nothing behind
        for (int r = 1; r < world; ++r) {
            const std::string an = ack_name(r);
            for (;;) {
                uint64_t got = 0;
                if (FILE *g = fopen(an.c_str(), "rb")) {
                    const size_t m = fread(&got, 1, sizeof got, g);
                    fclose(g);
                    if (m == sizeof got && got == nonce) break;
                }
                if (waited() > timeout_s) {
                    (void)unlink(path);
                    return amt_fail(AMT_ERR_COMM, "rank %d did not pick up %s within %.0f s", r, path, timeout_s);
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
            (void)unlink(an.c_str());
        }
        (void)unlink(path);
        return AMT_OK;
    }
    bool saw_stale = false;
    for (;;) {
        if (FILE *f = fopen(path, "rb")) {
            AmtRendezvousHeader hd;
            char id[AMT_UNIQUE_ID_BYTES];
            const size_t n = fread(&hd, 1, sizeof hd, f) + fread(id, 1, sizeof id, f);
            fclose(f);
            if (n == sizeof hd + sizeof id && memcmp(hd.magic, kRvMagic, 8) == 0) {
                if (hd.nonce == nonce) {
                    memcpy(id_out, id, sizeof id);
                    const std::string an = ack_name(rank), tmp = an + ".tmp";
                    FILE *g = fopen(tmp.c_str(), "wb");
                    if (!g) return amt_fail(AMT_ERR_COMM, "cannot write %s", tmp.c_str());
                    const size_t m = fwrite(&nonce, 1, sizeof nonce, g);
                    fclose(g);
                    if (m != sizeof nonce || rename(tmp.c_str(), an.c_str()) != 0)
                        return amt_fail(AMT_ERR_COMM, "cannot acknowledge %s", path);
                    return AMT_OK;
                }
                saw_stale = true;          // another launch's file: rank 0 of this one will replace it
            }
        }
        if (waited() > timeout_s)
            return amt_fail(AMT_ERR_COMM, saw_stale ? "%s belongs to another launch (nonce mismatch) after %.0f s; ranks that do not share "
                                                      "a parent process or a scheduler job id need the same AMT_RENDEZVOUS_NONCE"
                                                    : "no rendezvous file %s after %.0f s", path, timeout_s);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
}

struct amt_slab {
    amt_domain *dom = nullptr;
    int rank = 0, world = 1;
    int below = -1, above = -1;          // neighbour ranks, -1 = none
    bool overlap = true;
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t inputs_final = nullptr, edges_done = nullptr, t0 = nullptr, t1 = nullptr;
    double *red = nullptr;               // one device double for amt_slab_barrier / amt_slab_max
    int skew_us = 0;                     // test hook: the neighbours' rows arrive this late (amt_slab_set_skew_us)
};

extern "C" int amt_slab_destroy(amt_slab *s)
{
    if (!s) return AMT_OK;
    DeviceScope scope(s->dom ? s->dom->device : 0);
    if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
    if (s->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s->comm);
    for (hipEvent_t e : {s->inputs_final, s->edges_done, s->t0, s->t1})
        if (e) (void)hipEventDestroy(e);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    if (s->red) (void)hipFree(s->red);
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
    if (e == hipSuccess) e = hipMalloc((void **)&s->red, sizeof(double));
    if (e != hipSuccess) {
        amt_slab_destroy(s);
        return amt_fail(AMT_ERR_HIP, "amt_slab_create: %s", hipGetErrorString(e));
    }
    if (comm_needed) {
        int rc = amt_rccl_load();
        if (rc) { amt_slab_destroy(s); return rc; }
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        ncclResult_t r = g_rccl.CommInitRank(&s->comm, world, id, rank);
        if (r != ncclSuccess) {
            s->comm = nullptr;
            amt_slab_destroy(s);
            return amt_fail(AMT_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        }
    }
    *out = s;
    return AMT_OK;
}

namespace {
// rows that cross a slab boundary: row jte+1 of these comes from the rank above (its row jts) ...
const int kHaloFromAbove[] = {AMT_F_V, AMT_F_V_1, AMT_F_T_1, AMT_F_MUV, AMT_F_MSFVX_INV};   // :143-144, :241
// ... and row jts-1 of t_1 from the rank below (its row jte), :242
const int kHaloFromBelow[] = {AMT_F_T_1};

int amt_slab_enqueue_exchange(amt_slab *s, hipStream_t stream)
{
    if (s->below < 0 && s->above < 0) return AMT_OK;
    amt_domain *d = s->dom;
    const size_t idim = d->ime - d->ims + 1, kdim = d->kme - d->kms + 1;
    const ncclDataType_t dt = d->dtype_bytes == 8 ? ncclDouble : ncclFloat;
    auto row = [&](int f, int j, size_t &count) -> void * {
        count = amt_field_rank(f) == 3 ? idim * kdim : idim;
        return static_cast<char *>(d->field[f]) + (size_t)(j - d->jms) * count * d->dtype_bytes;
    };
    size_t n = 0;
    // per pair of ranks the order of sends matches the order of receives on the other side.
    // A failing call must not leave the group open (every later RCCL call of this thread would
    // be queued into it, the communicator's destruction included): remember the first error and
    // always close the group.
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    auto note = [&](ncclResult_t r, const char *w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; what = w; } };
    AMT_NCCL(g_rccl.GroupStart());
    if (s->below >= 0)
        for (int f : kHaloFromAbove) { void *q = row(f, d->jts, n); note(g_rccl.Send(q, n, dt, s->below, s->comm, stream), "ncclSend"); }
    if (s->above >= 0)
        for (int f : kHaloFromBelow) { void *q = row(f, d->jte, n); note(g_rccl.Send(q, n, dt, s->above, s->comm, stream), "ncclSend"); }
    if (s->above >= 0)
        for (int f : kHaloFromAbove) { void *q = row(f, d->jte + 1, n); note(g_rccl.Recv(q, n, dt, s->above, s->comm, stream), "ncclRecv"); }
    if (s->below >= 0)
        for (int f : kHaloFromBelow) { void *q = row(f, d->jts - 1, n); note(g_rccl.Recv(q, n, dt, s->below, s->comm, stream), "ncclRecv"); }
    note(g_rccl.GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess)
        return amt_fail(AMT_ERR_COMM, "%s failed in the halo exchange: %s", what,
                        g_rccl.GetErrorString ? g_rccl.GetErrorString(first) : "?");
    return AMT_OK;
}

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
        if (s->overlap) {
            AMT_HIP(hipEventRecord(s->inputs_final, d->stream));          // this sub-step's inputs are final
            AMT_HIP(hipStreamWaitEvent(s->comm_stream, s->inputs_final, 0));
            rc = amt_slab_tile<T>(s, d->stream, in_lo, in_hi, true);        // interior overlaps the exchange
            if (rc) { join(); return rc; }
        }
        if (s->skew_us > 0) {
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
        rc = amt_slab_enqueue_exchange(s, edge_stream);
        if (rc) { join(); return rc; }
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
    int rc = amt_slab_enqueue_exchange(s, s->comm_stream);
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
    return AMT_OK;
}

extern "C" int amt_slab_sync(amt_slab *s)
{
    if (!s) return amt_fail(AMT_ERR_INVALID_ARG, "null slab");
    DeviceScope scope(s->dom->device);
    AMT_HIP(hipStreamSynchronize(s->comm_stream));
    AMT_HIP(hipStreamSynchronize(s->dom->stream));
    return AMT_OK;
}

extern "C" long amt_slab_halo_bytes(const amt_slab *s)
{
    if (!s) return 0;
    const amt_domain *d = s->dom;
    const size_t idim = d->ime - d->ims + 1, kdim = d->kme - d->kms + 1;
    size_t per_pair = 0;
    for (int f : kHaloFromAbove) per_pair += amt_field_rank(f) == 3 ? idim * kdim : idim;
    for (int f : kHaloFromBelow) per_pair += amt_field_rank(f) == 3 ? idim * kdim : idim;
    return (long)(per_pair * d->dtype_bytes * ((s->below >= 0) + (s->above >= 0)));
}


// What the communicator itself says about this rank (ncclCommUserRank / ncclCommCount): a bench
// line can then show that `world` ranks really joined.  Without a communicator: 0 of 1.
extern "C" int amt_slab_comm_info(const amt_slab *s, int *rank, int *world)
{
    if (!s) return amt_fail(AMT_ERR_INVALID_ARG, "null slab");
    int r = 0, w = 1;
    if (s->comm) {
        AMT_NCCL(g_rccl.CommUserRank(s->comm, &r));
        AMT_NCCL(g_rccl.CommCount(s->comm, &w));
    }
    if (rank) *rank = r;
    if (world) *world = w;
    return AMT_OK;
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
    if (!s->comm || s->world == 1) return AMT_OK;
    AMT_HIP(hipMemcpyAsync(s->red, x, sizeof(double), hipMemcpyHostToDevice, s->comm_stream));
    AMT_NCCL(g_rccl.AllReduce(s->red, s->red, 1, ncclDouble, ncclMax, s->comm, s->comm_stream));
    AMT_HIP(hipMemcpyAsync(x, s->red, sizeof(double), hipMemcpyDeviceToHost, s->comm_stream));
    AMT_HIP(hipStreamSynchronize(s->comm_stream));
    return AMT_OK;
}

extern "C" int amt_slab_barrier(amt_slab *s)
{
    double zero = 0.0;
    return amt_slab_max(s, &zero);
}
