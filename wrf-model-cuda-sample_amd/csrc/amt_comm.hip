// amt_comm.hip -- what the multi-rank steppers share: the RCCL loader, the communicator id and its file rendezvous
// (include/amt_advance_mu_t.h section 5) and the halo-exchange engine with its two transports (amt_comm.h).
#include "amt_comm.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

// ---------------------------------------------------------------------------
// RCCL is opened with dlopen on first use: the library has no link-time dependency on it and
// single-GPU users never load it.
// ---------------------------------------------------------------------------
namespace {
struct AmtRccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitRankConfig)(ncclComm_t *, int, ncclUniqueId, int, ncclConfig_t *) = nullptr;   // optional
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
    r.CommInitRankConfig = reinterpret_cast<decltype(r.CommInitRankConfig)>(dlsym(lib, "ncclCommInitRankConfig"));
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

// Ids made without RCCL (it could not be loaded, or AMT_SLAB_TRANSPORT=ipc asks for none of it) start with this word:
// they name a launch for the IPC transport and are refused by the RCCL one.
static const char kLocalIdMagic[8] = {'A', 'M', 'T', 'L', 'O', 'C', 'I', 'D'};

static bool amt_env_transport_is_ipc()
{
    const char *e = getenv("AMT_SLAB_TRANSPORT");
    return e && !strcmp(e, "ipc");
}

extern "C" int amt_comm_unique_id(void *id_out)
{
    if (!id_out) return amt_fail(AMT_ERR_INVALID_ARG, "null id buffer");
    if (!amt_env_transport_is_ipc() && amt_rccl_load() == AMT_OK) {
        ncclUniqueId id;
        AMT_NCCL(g_rccl.GetUniqueId(&id));
        memcpy(id_out, &id, sizeof id);
        return AMT_OK;
    }
    // no RCCL: 120 random bytes behind the magic word (enough to name one launch on one node)
    unsigned char *q = static_cast<unsigned char *>(id_out);
    memcpy(q, kLocalIdMagic, 8);
    size_t got = 0;
    if (FILE *f = fopen("/dev/urandom", "rb")) {
        got = fread(q + 8, 1, AMT_UNIQUE_ID_BYTES - 8, f);
        fclose(f);
    }
    if (got != AMT_UNIQUE_ID_BYTES - 8) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        uint64_t h = amt_fnv1a(1469598103934665603ull, &ts, sizeof ts);
        const long pid = (long)getpid();
        for (size_t k = 8; k < AMT_UNIQUE_ID_BYTES; k += 8) {
            h = amt_fnv1a(h, &pid, sizeof pid);
            memcpy(q + k, &h, 8);
        }
    }
    return AMT_OK;
}

// Rendezvous for hosts without MPI: rank 0 creates the id and publishes it as `path` (written
// under a temporary name, then renamed) behind a header that carries the LAUNCH NONCE; the other
// ranks wait for a file whose nonce is theirs, read the id and acknowledge with `path.ack.<rank>`;
// rank 0 waits for the world-1 acknowledgements and removes every file.  A file left behind by an
// earlier launch (a crashed one: a completed one leaves nothing) carries another nonce and is
// never taken for this launch's, however recently it was written.

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
// several launcher agents (torchrun per node) make up this launch: the ranks do not share a parent process
static bool amt_launch_spans_agents()
{
    const char *lw = getenv("LOCAL_WORLD_SIZE"), *w = getenv("WORLD_SIZE"), *el = getenv("TORCHELASTIC_RUN_ID");
    return el && *el && lw && *lw && w && *w && atol(lw) != atol(w);
}

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
    // unless the scheduler's ids are all there is to agree on (its ranks have one daemon per node as parents) -- or the
    // local launcher is one of SEVERAL (one torchrun agent per node under a scheduler, LOCAL_WORLD_SIZE < WORLD_SIZE:
    // the ranks of the other nodes have other parents; what all nodes share is the job id, the run id and the master's
    // address)
    if (amt_launch_spans_agents()) mix_env("MASTER_ADDR");
    else if (!have_job || local_launcher) mix_parent();
    mix_env("MASTER_PORT");
    return h ? h : 1;
}

extern "C" int amt_comm_rendezvous_file(const char *path, uint64_t nonce, int rank, int world,
                                        double timeout_s, void *id_out)
{
    if (!path || !*path || !id_out || rank < 0 || world < 1 || rank >= world)
        return amt_fail(AMT_ERR_INVALID_ARG, "bad rendezvous argument");
    if (nonce == 0) {
        // several launcher agents and nothing all of them share (no scheduler job id, no explicit nonce): their ranks would
        // compute different nonces and time out on each other's file -- say so now
        static const char *const shared[] = {"AMT_RENDEZVOUS_NONCE", "SLURM_JOB_ID", "PMI_JOBID", "PMI_ID_JOB", "PMIX_NAMESPACE",
                                             "OMPI_MCA_ess_base_jobid", "LSB_JOBID", "PBS_JOBID"};
        bool any = false;
        for (const char *n : shared)
            if (const char *t = getenv(n); t && *t) any = true;
        if (!any && amt_launch_spans_agents() && !(getenv("TORCHELASTIC_RUN_ID") && strcmp(getenv("TORCHELASTIC_RUN_ID"), "none")))
            return amt_fail(AMT_ERR_COMM, "this launch spans several launcher agents (LOCAL_WORLD_SIZE %s of WORLD_SIZE %s) without a scheduler "
                                          "job id or a run id they share: export the same AMT_RENDEZVOUS_NONCE to every rank",
                            getenv("LOCAL_WORLD_SIZE"), getenv("WORLD_SIZE"));
        nonce = amt_comm_launch_nonce();
    }
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

// ---------------------------------------------------------------------------
// the exchange engine (amt_comm.h)
// ---------------------------------------------------------------------------
namespace {
constexpr int kMaxExports = 16;          // send segments of one rank (a grid patch has 12)
constexpr int kMaxPeers = 4;             // below, above, left, right
constexpr uint64_t kShmMagic = 0x414d54584348470aull;       // "AMTXCHG\n"

struct ShmExport {
    hipIpcMemHandle_t handle;            // of the sender's staging buffer (one allocation per rank)
    uint64_t offset;                     // of the segment's copy inside it
    uint64_t bytes;
    int32_t dest;
    int32_t pad;
};

// Everything two ranks share, in one POSIX shared-memory block every rank registers with hipHostRegister: the device
// kernels below poll and post the sequence numbers through it (fine-grained host memory: coherent for both processes
// and for every GPU of the node), the hosts use the header for the set-up rendezvous, the barrier and the max.
struct ShmHeader {
    uint64_t magic;
    uint32_t world;
    uint32_t rank_bytes;
    std::atomic<uint32_t> attached;      // ranks that have opened every handle they need
    std::atomic<uint32_t> bar_count;
    std::atomic<uint32_t> bar_gen;
    uint32_t pad[9];
};
struct ShmRank {
    std::atomic<uint32_t> published;     // 1: device, pid and exports[] are valid
    int32_t pid;
    int32_t nexports;
    int32_t device;
    char bus_id[32];
    ShmExport exports[kMaxExports];
    alignas(64) double red;              // amt_exchange_max
    alignas(64) unsigned long long ready;        // posted by this rank's device: its send segments are final for exchange n
    alignas(64) unsigned int error;              // set by this rank's own device-side waits when they give up
    alignas(64) unsigned long long pulled[1];    // [world]: pulled[p] is posted by rank p's device: p has pulled exchange n
};

static_assert(sizeof(ShmHeader) == 64, "one line");

struct AmtDevPtrs {
    unsigned long long *p[kMaxPeers];
    int n;
};

// "my send segments are final for exchange n", then wait until every source's are.  One wave; lane l polls source l.
__global__ void amt_xchg_post_and_wait(unsigned long long *mine, AmtDevPtrs src, unsigned long long n,
                                       unsigned long long ticks, unsigned long long skew_ticks, unsigned int *err)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) __hip_atomic_store(mine, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < src.n) {
        while (__hip_atomic_load(src.p[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < n) {
            if (wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    while (wall_clock64() - t0 < skew_ticks) __builtin_amdgcn_s_sleep(8);            // test hook: the neighbours' rows are this late
}
__global__ void amt_xchg_post(AmtDevPtrs dst, unsigned long long n)
{
    if ((int)threadIdx.x < dst.n) __hip_atomic_store(dst.p[threadIdx.x], n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void amt_xchg_wait(AmtDevPtrs src, unsigned long long n, unsigned long long ticks, unsigned int *err)
{
    if ((int)threadIdx.x < src.n) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(src.p[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < n) {
            if (wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
}

typedef unsigned int amt_v4u __attribute__((ext_vector_type(4)));
// One segment, by the `nthreads` threads of which this is number `tid`: 16 bytes per lane where both ends are 16-byte aligned,
// else 4 bytes per lane (a row of the caller's arrays is only element-aligned: fp32 rows of an odd element count start on every
// 4-byte boundary), the byte tail by the first lanes.  fill = true: the destination gets all-ones bytes instead (NaN in fp32
// and fp64): what a receive segment holds when its source never posted.
__device__ inline void amt_copy_segment(const void *src_, void *dst_, unsigned long long bytes, size_t tid, size_t nthreads, bool fill = false)
{
    const unsigned char *src = static_cast<const unsigned char *>(src_);
    unsigned char *dst = static_cast<unsigned char *>(dst_);
    size_t done;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const amt_v4u *s16 = reinterpret_cast<const amt_v4u *>(src);
        amt_v4u *d16 = reinterpret_cast<amt_v4u *>(dst);
        const amt_v4u ones = {~0u, ~0u, ~0u, ~0u};
        for (size_t e = tid; e < n16; e += nthreads) __builtin_nontemporal_store(fill ? ones : __builtin_nontemporal_load(s16 + e), d16 + e);
        done = n16 * 16;
    } else if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0) {
        const size_t n4 = bytes / 4;
        const unsigned int *s4 = reinterpret_cast<const unsigned int *>(src);
        unsigned int *d4 = reinterpret_cast<unsigned int *>(dst);
        for (size_t e = tid; e < n4; e += nthreads) __builtin_nontemporal_store(fill ? ~0u : __builtin_nontemporal_load(s4 + e), d4 + e);
        done = n4 * 4;
    } else {
        done = 0;
    }
    for (size_t e = done + tid; e < bytes; e += nthreads) dst[e] = fill ? (unsigned char)0xff : src[e];
}
struct AmtPullSegs {
    const void *src[kMaxExports];
    void *dst[kMaxExports];
    unsigned long long bytes[kMaxExports];
};
// The pull alone (host-waited exchange: the host has seen every source's "rows final" before this is enqueued): blockIdx.y =
// segment, 16 bytes per lane.  It runs after the interior, with the chip to itself: many workgroups, a few microseconds.
__global__ __launch_bounds__(256) void amt_xchg_pull(AmtPullSegs g)
{
    amt_copy_segment(g.src[blockIdx.y], g.dst[blockIdx.y], g.bytes[blockIdx.y], blockIdx.x * (size_t)blockDim.x + threadIdx.x,
                     (size_t)gridDim.x * blockDim.x);
}

// The whole of phase A as ONE kernel (the default between ranks that share a device; AMT_IPC_PULL=kernel elsewhere): a march
// workgroup takes a compute unit whole, so beside a slab's interior every kernel of the communication stream starts only where
// an interior workgroup ends -- a chain of wait, six copies and post would queue at one such place after the other
// (profiles/r05_slab_ab.md).  This kernel is enqueued BEFORE the interior, holds its `gridDim.x` compute units from the start of
// the sweep (RCCL's send/recv kernel holds 31), posts "rows final", waits for the sources, pulls every segment and posts "pulled".
struct AmtFusedArgs {
    unsigned long long *ready_mine;
    AmtDevPtrs src_ready, pulled_dst;
    unsigned long long n, ticks, skew_ticks;
    unsigned int *err, *wg_done;
    int nseg;
    AmtPullSegs segs;
};
__global__ __launch_bounds__(512) void amt_xchg_fused(AmtFusedArgs a)
{
    __shared__ int gave_up;
    if (threadIdx.x == 0) {
        gave_up = 0;
        if (blockIdx.x == 0) __hip_atomic_store(a.ready_mine, a.n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        for (int q = 0; q < a.src_ready.n; ++q)
            while (__hip_atomic_load(a.src_ready.p[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.n) {
                if (wall_clock64() - t0 > a.ticks) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    gave_up = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        while (wall_clock64() - t0 < a.skew_ticks) __builtin_amdgcn_s_sleep(8);      // test hook: the neighbours' rows are this late
    }
    __syncthreads();
    // a source that never posted: its rows are not valid -- the receive segments get NaN (all-ones) instead of whatever the
    // peer's buffer holds, so that the sweep's results are loudly wrong as well as reported (amt_exchange_check)
    const bool poison = gave_up != 0;
    for (int g = 0; g < a.nseg; ++g)
        amt_copy_segment(a.segs.src[g], a.segs.dst[g], a.segs.bytes[g], blockIdx.x * (size_t)blockDim.x + threadIdx.x,
                         (size_t)gridDim.x * blockDim.x, poison);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(a.wg_done, 1u) == gridDim.x - 1) {              // the last workgroup to finish posts
        *a.wg_done = 0;
        for (int q = 0; q < a.pulled_dst.n; ++q) __hip_atomic_store(a.pulled_dst.p[q], a.n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Kind of the copy-engine pull.  Between GPUs the runtime moves a device-to-device copy with SDMA by itself.  On ONE device it
// is a blit kernel, which needs a compute unit; hipMemcpyDeviceToDeviceNoCU ("without using compute units") forces the engine
// there too -- measured on this pool (profiles/r05_slab_ab.md): the kernel trace then shows no copy kernel at all, but the six
// 2 MiB copies of a slab interface take ~80 us each through the engine against 3 us as blit kernels, and the loopback sweep
// is 8 % slower.  So: the plain kind, and AMT_IPC_ENGINE_NOCU=1 for whoever wants the engine by name.
hipMemcpyKind amt_engine_copy_kind()
{
    static const bool nocu = [] { const char *e = getenv("AMT_IPC_ENGINE_NOCU"); return e && *e && atoi(e) != 0; }();
    return nocu ? hipMemcpyDeviceToDeviceNoCU : hipMemcpyDeviceToDevice;
}

double amt_env_seconds(const char *name, double dflt)
{
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const double v = atof(e);
    return v > 0 ? v : dflt;
}

std::mutex g_xchg_mutex;
// (unique id, rank) -> exchanges that rank has created with the id.  Per RANK, not per process: ranks that live in one process
// (threads) count separately and agree with ranks elsewhere; a create that fails does not advance it.
std::map<std::pair<uint64_t, int>, unsigned> g_xchg_instances;
}  // namespace

struct AmtExchange {
    int transport = AMT_XCHG_RCCL;
    int rank = 0, world = 1, device = 0;
    bool self_loop = false;
    std::vector<AmtSeg> sends, recvs;
    // RCCL
    ncclComm_t comm = nullptr;
    double *red = nullptr;
    // IPC
    void *shm = nullptr;
    size_t shm_bytes = 0;
    bool shm_registered = false;
    char *shm_dev = nullptr;                 // the block as the device addresses it
    void *stage = nullptr;                   // copies of the send segments: what the neighbours map and pull
    std::vector<size_t> stage_off;
    std::vector<void *> opened;              // hipIpcOpenMemHandle mappings to close
    std::vector<const void *> recv_src;      // per receive segment: where to pull it from
    std::vector<int> sources, dests;         // distinct peer ranks
    unsigned long long seq = 0;              // exchanges enqueued so far
    unsigned long long exchanges = 0;        // phase A starts of either transport (what AMT_TEST_FAULT counts)
    unsigned long long released = 0;
    bool pull_kernel = false;               // one fused kernel (wait + pull + post) instead of wait kernel, copy-engine pulls, post kernel
    bool same_device_peers = true;          // every source rank computes on this rank's device (ranks that share a GPU; loopback)
    int pull_wgs = 4;
    unsigned int *wg_done = nullptr;        // device counter of the fused kernel
    unsigned long long skew_ticks = 0;      // test hook (amt_exchange_set_skew_us)
    unsigned long long ticks = 0;
    double host_timeout = 120.0;

    ShmHeader *hdr() const { return static_cast<ShmHeader *>(shm); }
    ShmRank *slot(int r) const { return reinterpret_cast<ShmRank *>(static_cast<char *>(shm) + sizeof(ShmHeader) + (size_t)r * hdr()->rank_bytes); }
    template <typename Q> Q *dev(Q *host_ptr) const { return reinterpret_cast<Q *>(shm_dev + (reinterpret_cast<char *>(host_ptr) - static_cast<char *>(shm))); }
};

namespace {
int amt_ipc_barrier(AmtExchange *x)
{
    ShmHeader *h = x->hdr();
    const uint32_t gen = h->bar_gen.load(std::memory_order_acquire);
    if (h->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)x->world) {
        h->bar_count.store(0, std::memory_order_relaxed);
        h->bar_gen.store(gen + 1, std::memory_order_release);
        return AMT_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (h->bar_gen.load(std::memory_order_acquire) == gen) {
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > x->host_timeout)
            return amt_fail(AMT_ERR_COMM, "IPC barrier: %u of %d ranks arrived within %.0f s", h->bar_count.load(), x->world, x->host_timeout);
    }
    return AMT_OK;
}

int amt_ipc_setup(AmtExchange *x, const void *unique_id)
{
    const int world = x->world, rank = x->rank;
    // AMT_IPC_DEBUG=1: one stderr line per step of the set-up, with the seconds since it began (field diagnosis of a launch
    // whose ranks do not find each other)
    static const bool debug = [] { const char *e = getenv("AMT_IPC_DEBUG"); return e && *e && atoi(e) != 0; }();
    const auto t_begin = std::chrono::steady_clock::now();
    auto trace = [&](const char *what) {
        if (debug)
            fprintf(stderr, "[amt ipc] rank %d/%d +%.3f s: %s\n", rank, world,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), what);
    };
    if ((int)x->sends.size() > kMaxExports || (int)x->recvs.size() > kMaxExports)
        return amt_fail(AMT_ERR_INVALID_ARG, "more than %d segments per rank", kMaxExports);
    auto distinct = [](const std::vector<AmtSeg> &v) {
        std::vector<int> p;
        for (const AmtSeg &s : v)
            if (std::find(p.begin(), p.end(), s.peer) == p.end()) p.push_back(s.peer);
        return p;
    };
    x->sources = distinct(x->recvs);
    x->dests = distinct(x->sends);
    if ((int)x->sources.size() > kMaxPeers || (int)x->dests.size() > kMaxPeers)
        return amt_fail(AMT_ERR_INVALID_ARG, "more than %d neighbours", kMaxPeers);
    x->host_timeout = amt_env_seconds("AMT_IPC_TIMEOUT_S", 120.0);
    x->ticks = (unsigned long long)(amt_env_seconds("AMT_IPC_DEVICE_TIMEOUT_S", 30.0) * 1e8);     // 100 MHz wall clock
    if (const char *e = getenv("AMT_IPC_PULL_WGS")) { const int n = atoi(e); if (n >= 1 && n <= 64) x->pull_wgs = n; }

    // the block: one name per (unique id, how many exchanges THIS RANK made with it before) -- every rank creates its
    // exchanges in the same order, so the counters agree.  Whoever comes first creates it (O_EXCL), sizes it and writes the
    // header, the magic word last; everybody else opens it, waits for the magic word and VALIDATES world and rank_bytes (two
    // launches, or two builds, that meet in one name must not talk).  The name goes when every rank is attached -- every
    // rank unlinks, the first one wins -- and on the creator's error paths; a create that fails does not advance the counter.
    uint64_t idh = amt_fnv1a(1469598103934665603ull, unique_id, AMT_UNIQUE_ID_BYTES);
    unsigned instance;
    {
        std::lock_guard<std::mutex> lk(g_xchg_mutex);
        instance = g_xchg_instances[{idh, rank}];
    }
    char name[80];
    snprintf(name, sizeof name, "/amt_xchg_%016llx_%u", (unsigned long long)idh, instance);
    const size_t rank_bytes = (sizeof(ShmRank) + (size_t)world * sizeof(unsigned long long) + 63) / 64 * 64;
    x->shm_bytes = (sizeof(ShmHeader) + rank_bytes * world + 4095) / 4096 * 4096;
    bool creator = true;
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 && errno == EEXIST) {
        creator = false;
        fd = shm_open(name, O_RDWR, 0600);
    }
    if (fd < 0) return amt_fail(AMT_ERR_COMM, "shm_open(%s) failed: %s", name, strerror(errno));
    struct Unlinker { const char *n; bool armed; ~Unlinker() { if (armed) (void)shm_unlink(n); } } unlinker{name, creator};
    const auto t_open = std::chrono::steady_clock::now();
    auto opened_for = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count(); };
    if (creator) {
        if (ftruncate(fd, (off_t)x->shm_bytes) != 0) { close(fd); return amt_fail(AMT_ERR_COMM, "ftruncate(%s) failed: %s", name, strerror(errno)); }
    } else {
        struct stat sb;                          // the creator sizes the block before anything else
        for (;;) {
            if (fstat(fd, &sb) != 0) { close(fd); return amt_fail(AMT_ERR_COMM, "fstat(%s) failed: %s", name, strerror(errno)); }
            if ((size_t)sb.st_size >= x->shm_bytes) break;
            if (sb.st_size != 0 || opened_for() > x->host_timeout) {
                close(fd);
                return amt_fail(AMT_ERR_COMM, "%s holds %lld bytes, this rank expects %zu (another launch, or another build, under the same name?)",
                                name, (long long)sb.st_size, x->shm_bytes);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void *m = mmap(nullptr, x->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return amt_fail(AMT_ERR_COMM, "mmap(%s) failed: %s", name, strerror(errno));
    x->shm = m;
    trace(name);
    ShmHeader *h = x->hdr();
    if (creator) {
        h->world = (uint32_t)world;
        h->rank_bytes = (uint32_t)rank_bytes;
        __atomic_store_n(&h->magic, kShmMagic, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) != kShmMagic) {
            if (opened_for() > x->host_timeout) return amt_fail(AMT_ERR_COMM, "%s was never initialised by the rank that created it (%.0f s)", name, x->host_timeout);
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (h->world != (uint32_t)world || h->rank_bytes != (uint32_t)rank_bytes)
            return amt_fail(AMT_ERR_COMM, "%s was set up for %u ranks of %u bytes, this rank is one of %d with %zu (ranks of different launches or builds)",
                            name, h->world, h->rank_bytes, world, rank_bytes);
    }
    AMT_HIP(hipHostRegister(m, x->shm_bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    x->shm_registered = true;
    void *dptr = nullptr;
    AMT_HIP(hipHostGetDevicePointer(&dptr, m, 0));
    x->shm_dev = static_cast<char *>(dptr);
    trace("block registered with the device");

    // publish my send segments
    ShmRank *me = x->slot(rank);
    me->pid = (int32_t)getpid();
    me->device = x->device;
    (void)hipDeviceGetPCIBusId(me->bus_id, (int)sizeof me->bus_id, x->device);
    me->nexports = (int32_t)x->sends.size();
    me->error = 0;
    // What a neighbour pulls is a STAGING buffer this exchange owns, not the caller's arrays: one small hipMalloc holds a copy
    // of every send segment (a kernel refreshes it per exchange, 8 MB per slab interface: microseconds), and only ITS handle
    // is traded.  The caller's arrays may then be anything a kernel can read -- hipMalloc of any size, a pooled allocator,
    // hipMemMap ranges -- and what the neighbour maps is 8 MB, not a 4 GiB array (hipIpcOpenMemHandle of an allocation of
    // 4 GiB or more never returned on this ROCm: both ranks sat in it, profiles/r05_slab_ab.md section 5).
    size_t stage_bytes = 0;
    x->stage_off.assign(x->sends.size(), 0);
    for (size_t k = 0; k < x->sends.size(); ++k) {
        x->stage_off[k] = stage_bytes;
        stage_bytes += (x->sends[k].bytes + 255) / 256 * 256;
    }
    hipIpcMemHandle_t stage_handle{};
    if (stage_bytes) {
        hipError_t he = hipMalloc(&x->stage, stage_bytes);
        if (he != hipSuccess) return amt_fail(AMT_ERR_ALLOC, "hipMalloc of the %zu-byte halo staging buffer failed: %s", stage_bytes, hipGetErrorString(he));
        if (!x->self_loop) {
            he = hipIpcGetMemHandle(&stage_handle, x->stage);
            if (he != hipSuccess) return amt_fail(AMT_ERR_COMM, "hipIpcGetMemHandle of the halo staging buffer failed: %s", hipGetErrorString(he));
        }
    }
    for (size_t k = 0; k < x->sends.size(); ++k) {
        ShmExport &e = me->exports[k];
        e.dest = x->sends[k].peer;
        e.bytes = x->sends[k].bytes;
        e.offset = x->stage_off[k];
        e.handle = stage_handle;
    }
    me->published.store(1, std::memory_order_release);
    trace("send segments published");

    // open what I receive: the k-th receive from p is the k-th segment p sends to me
    const auto t0 = std::chrono::steady_clock::now();
    auto waited = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    x->recv_src.assign(x->recvs.size(), nullptr);
    std::map<std::string, void *> mapped;        // one mapping per allocation
    for (int p : x->sources) {
        if (p < 0 || p >= world) return amt_fail(AMT_ERR_INVALID_ARG, "receive from rank %d of %d", p, world);
        ShmRank *src = x->slot(p);
        while (!src->published.load(std::memory_order_acquire)) {
            if (waited() > x->host_timeout)
                return amt_fail(AMT_ERR_COMM, "rank %d did not publish its halo rows in %s within %.0f s (AMT_IPC_TIMEOUT_S)", p, name, x->host_timeout);
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        int nth = 0;
        for (size_t r = 0; r < x->recvs.size(); ++r) {
            if (x->recvs[r].peer != p) continue;
            const int kth = nth++;
            int seen = 0, found = -1;
            for (int k = 0; k < src->nexports; ++k)
                if (src->exports[k].dest == rank && seen++ == kth) { found = k; break; }
            if (found < 0) return amt_fail(AMT_ERR_COMM, "rank %d sends fewer segments to rank %d than it expects", p, rank);
            const ShmExport &e = src->exports[found];
            if (e.bytes != x->recvs[r].bytes)
                return amt_fail(AMT_ERR_COMM, "segment sizes differ between rank %d (%llu bytes) and rank %d (%zu bytes)", p,
                                (unsigned long long)e.bytes, rank, x->recvs[r].bytes);
            if (x->self_loop) {                       // the rank's own staging buffer serves: nothing to open
                x->recv_src[r] = static_cast<char *>(x->stage) + e.offset;
                continue;
            }
            const std::string key(reinterpret_cast<const char *>(&e.handle), sizeof e.handle);
            auto it = mapped.find(key);
            if (it == mapped.end()) {
                void *q = nullptr;
                hipError_t he = hipIpcOpenMemHandle(&q, e.handle, hipIpcMemLazyEnablePeerAccess);
                if (he != hipSuccess)
                    return amt_fail(AMT_ERR_COMM, "hipIpcOpenMemHandle of rank %d's rows failed: %s (pid %d, device %s; "
                                                  "HSA_ENABLE_IPC_MODE_LEGACY=%s)", p, hipGetErrorString(he), (int)src->pid, src->bus_id,
                                    getenv("HSA_ENABLE_IPC_MODE_LEGACY") ? getenv("HSA_ENABLE_IPC_MODE_LEGACY") : "unset");
                x->opened.push_back(q);
                it = mapped.emplace(key, q).first;
                trace("opened a peer allocation");
            }
            x->recv_src[r] = static_cast<char *>(it->second) + e.offset;
        }
    }
    // How the rows are pulled.  Between GPUs: the copy engine (SDMA over xGMI, no compute unit).  Between ranks that share a
    // device -- and in loopback -- a peer "copy" is a blit kernel anyway (22 us each through the IPC mapping, profiles/
    // r05_ipc_probe.txt): there one fused kernel does the whole phase.  AMT_IPC_PULL=engine|kernel overrides.
    for (int p : x->sources)
        if (strncmp(x->slot(p)->bus_id, me->bus_id, sizeof me->bus_id) != 0) x->same_device_peers = false;
    x->pull_kernel = x->same_device_peers;
    if (const char *e = getenv("AMT_IPC_PULL")) {
        if (!strcmp(e, "kernel")) x->pull_kernel = true;
        else if (!strcmp(e, "engine")) x->pull_kernel = false;
        else if (*e) return amt_fail(AMT_ERR_INVALID_ARG, "AMT_IPC_PULL must be engine or kernel, not '%s'", e);
    }
    AMT_HIP(hipMalloc((void **)&x->wg_done, sizeof(unsigned int)));
    AMT_HIP(hipMemset(x->wg_done, 0, sizeof(unsigned int)));
    // everybody attached: the name can go (the mappings stay)
    trace("attached; waiting for the others");
    h->attached.fetch_add(1, std::memory_order_acq_rel);
    while (h->attached.load(std::memory_order_acquire) < (uint32_t)world) {
        if (waited() > x->host_timeout)
            return amt_fail(AMT_ERR_COMM, "%u of %d ranks attached to %s within %.0f s", h->attached.load(), world, name, x->host_timeout);
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    (void)shm_unlink(name);                      // every rank: the first one removes the name, the mappings stay
    unlinker.armed = false;
    {
        std::lock_guard<std::mutex> lk(g_xchg_mutex);
        ++g_xchg_instances[{idh, rank}];         // only a set-up that succeeded counts
    }
    return AMT_OK;
}

int amt_rccl_exchange(AmtExchange *x, hipStream_t stream)
{
    // per pair of ranks the order of sends matches the order of receives on the other side.  A failing call must not
    // leave the group open (every later RCCL call of this thread would be queued into it, the communicator's destruction
    // included): remember the first error and always close the group.
    if (amt_test_fault("skip_group", x->exchanges)) return AMT_OK;      // fault injection: every rank skips the same group
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    auto note = [&](ncclResult_t r, const char *w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; what = w; } };
    AMT_NCCL(g_rccl.GroupStart());
    for (const AmtSeg &s : x->sends) note(g_rccl.Send(s.ptr, s.bytes, ncclChar, s.peer, x->comm, stream), "ncclSend");
    for (const AmtSeg &s : x->recvs) note(g_rccl.Recv(s.ptr, s.bytes, ncclChar, s.peer, x->comm, stream), "ncclRecv");
    note(g_rccl.GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess)
        return amt_fail(AMT_ERR_COMM, "%s failed in the halo exchange: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(first) : "?");
    return AMT_OK;
}
}  // namespace

int amt_exchange_destroy(AmtExchange *x)
{
    if (!x) return AMT_OK;
    DeviceScope scope(x->device);
    if (x->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(x->comm);
    if (x->red) (void)hipFree(x->red);
    if (x->wg_done) (void)hipFree(x->wg_done);
    if (x->stage) (void)hipFree(x->stage);
    for (void *q : x->opened) (void)hipIpcCloseMemHandle(q);
    if (x->shm_registered) (void)hipHostUnregister(x->shm);
    if (x->shm) (void)munmap(x->shm, x->shm_bytes);
    delete x;
    return AMT_OK;
}

int amt_exchange_create(AmtExchange **out, int transport, int rank, int world, const void *unique_id, int device,
                        const AmtSeg *sends, int nsend, const AmtSeg *recvs, int nrecv, bool self_loop)
{
    *out = nullptr;
    if (transport != AMT_XCHG_RCCL && transport != AMT_XCHG_IPC) return amt_fail(AMT_ERR_INVALID_ARG, "unknown halo transport %d", transport);
    AmtExchange *x = new (std::nothrow) AmtExchange;
    if (!x) return amt_fail(AMT_ERR_ALLOC, "host allocation failed");
    x->transport = transport; x->rank = rank; x->world = world; x->device = device; x->self_loop = self_loop;
    x->sends.assign(sends, sends + nsend);
    x->recvs.assign(recvs, recvs + nrecv);
    const bool needed = world > 1 || self_loop;
    if (!needed) { *out = x; return AMT_OK; }
    if (!unique_id) { delete x; return amt_fail(AMT_ERR_INVALID_ARG, "a communicator needs the unique id"); }
    DeviceScope scope(device);
    int rc = AMT_OK;
    if (transport == AMT_XCHG_RCCL) {
        rc = amt_rccl_load();
        if (rc == AMT_OK && !memcmp(unique_id, kLocalIdMagic, 8))
            rc = amt_fail(AMT_ERR_COMM, "this communicator id was made without RCCL (AMT_SLAB_TRANSPORT=ipc, or librccl was not found): "
                                        "it serves the IPC transport only");
        if (rc == AMT_OK && hipMalloc((void **)&x->red, sizeof(double)) != hipSuccess) rc = amt_fail(AMT_ERR_ALLOC, "device allocation failed");
        if (rc == AMT_OK) {
            ncclUniqueId id;
            memcpy(&id, unique_id, sizeof id);
            // The send/recv kernel of a communicator runs one workgroup per channel and every one of them holds a compute unit
            // for as long as the kernel waits for the wire: 31 by default, beside an interior launch whose workgroups need whole
            // units (profiles/r05_slab_ab.md).  Six rows per neighbour do not need 31 channels: this communicator is capped at
            // AMT_RCCL_MAX_CTAS (4; 0 = RCCL's own default).
            int max_ctas = 4;
            if (const char *e = getenv("AMT_RCCL_MAX_CTAS")) max_ctas = atoi(e);
            ncclResult_t r;
            if (max_ctas > 0 && g_rccl.CommInitRankConfig) {
                ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
                cfg.maxCTAs = max_ctas;
                cfg.minCTAs = 1;
                r = g_rccl.CommInitRankConfig(&x->comm, world, id, rank, &cfg);
            } else {
                r = g_rccl.CommInitRank(&x->comm, world, id, rank);
            }
            if (r != ncclSuccess) {
                x->comm = nullptr;
                rc = amt_fail(AMT_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
            }
        }
    } else {
        rc = amt_ipc_setup(x, unique_id);
    }
    if (rc != AMT_OK) {
        const std::string keep = amt_last_error();       // destroy must not lose the diagnosis
        amt_exchange_destroy(x);
        return amt_fail(rc, "%s", keep.c_str());
    }
    *out = x;
    return AMT_OK;
}

void amt_exchange_set_skew_us(AmtExchange *x, int microseconds) { if (x) x->skew_ticks = (unsigned long long)microseconds * 100ull; }
// does the transport carry the test skew itself (IPC: inside its waiting kernel)?  RCCL: the stepper delays the stream
bool amt_exchange_owns_skew(const AmtExchange *x) { return x && x->transport == AMT_XCHG_IPC && x->shm; }
const char *amt_exchange_pull_mode(const AmtExchange *x)
{
    if (!x || x->transport != AMT_XCHG_IPC || !x->shm) return "";
    return x->pull_kernel ? "fused kernel" : "copy engine";
}
bool amt_exchange_active(const AmtExchange *x) { return x && (!x->sends.empty() || !x->recvs.empty()) && (x->comm || x->shm); }
int amt_exchange_transport(const AmtExchange *x) { return x ? x->transport : AMT_XCHG_RCCL; }
void amt_exchange_bytes(const AmtExchange *x, size_t *sent, size_t *received)
{
    *sent = *received = 0;
    if (!x) return;
    for (const AmtSeg &s : x->sends) *sent += s.bytes;
    for (const AmtSeg &s : x->recvs) *received += s.bytes;
}

// workgroups per segment of the plain copy kernels (staging refresh, host-waited pull): both run with the chip to themselves (in
// front of / behind the interior), and a 2 MB row per segment wants more than a handful -- 16 / 32 per segment (r05) against 64:
// one rank of 8 in loopback 1.920 -> 1.909 ms per sweep, a 1024 x 1024 patch +19.0 -> +17.9 % (r06).  AMT_IPC_COPY_WGS overrides.
static unsigned amt_copy_wgs(unsigned dflt)
{
    static const int env = [] { const char *e = getenv("AMT_IPC_COPY_WGS"); return e && *e ? atoi(e) : 0; }();
    return env >= 1 && env <= 1024 ? (unsigned)env : dflt;
}

// IPC: refresh the staging copies of the send segments (the stream must be one on which the segments are final)
static int amt_ipc_stage(AmtExchange *x, hipStream_t stream)
{
    if (x->sends.empty() || amt_test_fault("skip_stage", x->exchanges)) return AMT_OK;
    AmtPullSegs g{};
    for (size_t k = 0; k < x->sends.size(); ++k) {
        g.src[k] = x->sends[k].ptr;
        g.dst[k] = static_cast<char *>(x->stage) + x->stage_off[k];
        g.bytes[k] = x->sends[k].bytes;
    }
    hipLaunchKernelGGL(amt_xchg_pull, dim3(amt_copy_wgs(64), (unsigned)x->sends.size()), dim3(256), 0, stream, g);
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}

int amt_exchange_enqueue(AmtExchange *x, hipStream_t stream, bool alone)
{
    if (!amt_exchange_active(x)) return AMT_OK;
    ++x->exchanges;
    if (x->transport == AMT_XCHG_RCCL) return amt_rccl_exchange(x, stream);
    if (int rc = amt_ipc_stage(x, stream)) return rc;
    const bool no_pull = amt_test_fault("skip_pull", x->exchanges);
    const unsigned long long n = ++x->seq;
    ShmRank *me = x->slot(x->rank);
    AmtDevPtrs src{}, done{};
    for (int p : x->sources) src.p[src.n++] = x->dev(&x->slot(p)->ready);
    for (int p : x->sources) done.p[done.n++] = x->dev(&x->slot(p)->pulled[x->rank]);
    if (x->pull_kernel) {
        AmtFusedArgs a{};
        a.ready_mine = x->dev(&me->ready); a.src_ready = src; a.pulled_dst = done;
        a.n = n; a.ticks = x->ticks; a.skew_ticks = x->skew_ticks;
        a.err = x->dev(&me->error); a.wg_done = x->wg_done;
        a.nseg = no_pull ? 0 : (int)x->recvs.size();
        for (size_t r = 0; r < x->recvs.size(); ++r) { a.segs.src[r] = x->recv_src[r]; a.segs.dst[r] = x->recvs[r].ptr; a.segs.bytes[r] = x->recvs[r].bytes; }
        // beside an interior every workgroup of this kernel holds a compute unit while it waits: few; alone on the chip: many
        hipLaunchKernelGGL(amt_xchg_fused, dim3((unsigned)(alone ? 64 : x->pull_wgs)), dim3(512), 0, stream, a);
        AMT_HIP(hipGetLastError());
        return AMT_OK;
    }
    hipLaunchKernelGGL(amt_xchg_post_and_wait, dim3(1), dim3(64), 0, stream, x->dev(&me->ready), src, n, x->ticks, x->skew_ticks, x->dev(&me->error));
    for (size_t r = 0; r < x->recvs.size() && !no_pull; ++r)
        AMT_HIP(hipMemcpyAsync(x->recvs[r].ptr, x->recv_src[r], x->recvs[r].bytes, amt_engine_copy_kind(), stream));
    if (done.n) hipLaunchKernelGGL(amt_xchg_post, dim3(1), dim3(64), 0, stream, done, n);
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}

// ---- the host-waited exchange (IPC): post on the domain's stream, wait on the HOST, pull behind the interior ------------------
// Phase A in three parts, so that NOTHING of the exchange holds a compute unit while the interior runs (a march workgroup
// takes a unit whole: a one-round interior has no unit to spare, profiles/r05_slab_ab.md):
//   amt_exchange_enqueue_post   "my send segments are final for exchange n" -- one wave on the stream that produced them
//   amt_exchange_host_wait      the calling HOST thread polls the mailbox until every source has posted n (or AMT_IPC_TIMEOUT_S)
//   amt_exchange_enqueue_pull   the rows, then "pulled n" to the sources
int amt_exchange_enqueue_post(AmtExchange *x, hipStream_t stream)
{
    if (!amt_exchange_active(x) || x->transport != AMT_XCHG_IPC) return amt_fail(AMT_ERR_INVALID_ARG, "the host-waited exchange needs the IPC transport");
    ++x->exchanges;
    if (int rc = amt_ipc_stage(x, stream)) return rc;
    const unsigned long long n = ++x->seq;
    AmtDevPtrs mine{};
    mine.p[mine.n++] = x->dev(&x->slot(x->rank)->ready);
    hipLaunchKernelGGL(amt_xchg_post, dim3(1), dim3(64), 0, stream, mine, n);
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}

int amt_exchange_host_wait(AmtExchange *x)
{
    if (!amt_exchange_active(x) || x->transport != AMT_XCHG_IPC) return AMT_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (int p : x->sources) {
        unsigned spins = 0;
        while (__atomic_load_n(&x->slot(p)->ready, __ATOMIC_ACQUIRE) < x->seq) {
            if (++spins > 4000) {
                std::this_thread::sleep_for(std::chrono::microseconds(20));
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > x->host_timeout)
                    return amt_fail(AMT_ERR_COMM, "IPC halo exchange %llu: rank %d did not post its rows within %.0f s (AMT_IPC_TIMEOUT_S)",
                                    x->seq, p, x->host_timeout);
            }
        }
    }
    if (x->skew_ticks) {                                  // test hook: the neighbours' rows are this late -- counted from the moment
        // the last source posted (in loopback: from the start of this rank's own sweep on the device)
        const auto late = std::chrono::steady_clock::now() + std::chrono::nanoseconds(x->skew_ticks * 10ull);
        while (std::chrono::steady_clock::now() < late) { }
    }
    return AMT_OK;
}

int amt_exchange_enqueue_pull(AmtExchange *x, hipStream_t stream)
{
    if (!amt_exchange_active(x) || x->transport != AMT_XCHG_IPC) return AMT_OK;
    if (amt_test_fault("skip_pull", x->exchanges)) {
        // fault injection: nothing is pulled, the protocol goes on
    } else if (x->pull_kernel && !x->recvs.empty()) {
        AmtPullSegs g{};
        for (size_t r = 0; r < x->recvs.size(); ++r) { g.src[r] = x->recv_src[r]; g.dst[r] = x->recvs[r].ptr; g.bytes[r] = x->recvs[r].bytes; }
        hipLaunchKernelGGL(amt_xchg_pull, dim3(amt_copy_wgs(64), (unsigned)x->recvs.size()), dim3(256), 0, stream, g);
    } else {
        for (size_t r = 0; r < x->recvs.size(); ++r)
            AMT_HIP(hipMemcpyAsync(x->recvs[r].ptr, x->recv_src[r], x->recvs[r].bytes, amt_engine_copy_kind(), stream));
    }
    AmtDevPtrs done{};
    for (int p : x->sources) done.p[done.n++] = x->dev(&x->slot(p)->pulled[x->rank]);
    if (done.n) hipLaunchKernelGGL(amt_xchg_post, dim3(1), dim3(64), 0, stream, done, x->seq);
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}

int amt_exchange_enqueue_release(AmtExchange *x, hipStream_t stream)
{
    if (!amt_exchange_active(x) || x->transport != AMT_XCHG_IPC || x->released == x->seq) return AMT_OK;
    ShmRank *me = x->slot(x->rank);
    AmtDevPtrs w{};
    for (int p : x->dests) w.p[w.n++] = x->dev(&me->pulled[p]);
    if (w.n) hipLaunchKernelGGL(amt_xchg_wait, dim3(1), dim3(64), 0, stream, w, x->seq, x->ticks, x->dev(&me->error));
    x->released = x->seq;
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}

int amt_exchange_check(AmtExchange *x)
{
    if (!x || !x->shm) return AMT_OK;
    const unsigned int e = __atomic_load_n(&x->slot(x->rank)->error, __ATOMIC_ACQUIRE);
    if (e)
        return amt_fail(AMT_ERR_COMM, "IPC halo exchange: a neighbour of rank %d did not %s within %.0f s (AMT_IPC_DEVICE_TIMEOUT_S); the halo "
                                      "rows of that sweep are not valid", x->rank, e == 1 ? "post its rows" : "pull this rank's rows", x->ticks / 1e8);
    return AMT_OK;
}

int amt_exchange_info(const AmtExchange *x, int *rank, int *world)
{
    int r = 0, w = 1;
    if (x && x->comm) {
        AMT_NCCL(g_rccl.CommUserRank(x->comm, &r));
        AMT_NCCL(g_rccl.CommCount(x->comm, &w));
    } else if (x && x->shm) {
        r = x->rank;
        w = (int)x->hdr()->attached.load();
    }
    if (rank) *rank = r;
    if (world) *world = w;
    return AMT_OK;
}

int amt_exchange_max(AmtExchange *x, double *v, hipStream_t stream)
{
    if (!x || x->world == 1) return AMT_OK;
    if (x->comm) {
        AMT_HIP(hipMemcpyAsync(x->red, v, sizeof(double), hipMemcpyHostToDevice, stream));
        AMT_NCCL(g_rccl.AllReduce(x->red, x->red, 1, ncclDouble, ncclMax, x->comm, stream));
        AMT_HIP(hipMemcpyAsync(v, x->red, sizeof(double), hipMemcpyDeviceToHost, stream));
        AMT_HIP(hipStreamSynchronize(stream));
        return AMT_OK;
    }
    if (!x->shm) return AMT_OK;
    x->slot(x->rank)->red = *v;
    int rc = amt_ipc_barrier(x);
    if (rc) return rc;
    double m = *v;
    for (int r = 0; r < x->world; ++r) m = x->slot(r)->red > m ? x->slot(r)->red : m;
    rc = amt_ipc_barrier(x);                         // nobody overwrites its value before all have read it
    *v = m;
    return rc;
}
