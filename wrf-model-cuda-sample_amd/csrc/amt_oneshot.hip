// amt_oneshot.hip -- the one-shot host drop-in amt_advance_mu_t_f32/_f64: host arrays in, host
// arrays out (the job of the reference's CUDA wrapper, advance_mu_t_no_async.cu:178-423: alloc,
// H2D, launch, D2H, free), streamed.
#include "amt_internal.h"
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/syscall.h>
#include <unistd.h>

namespace {
int amt_env_flag(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
// What a one-shot call needs on the device besides the data: three streams, six events and a
// buffer arena.  Creating and destroying them costs ~10 ms per call -- more than the whole call
// at WRF patch sizes -- so each host thread keeps its own set between calls (per thread: WRF
// calls advance_mu_t from OpenMP tile threads).  amt_host_release() frees the calling thread's.
struct HostWorkspace {
    int device = -1;
    hipStream_t up = nullptr, comp = nullptr, down = nullptr;
    hipEvent_t uploaded[2] = {}, computed[2] = {}, drained[2] = {};
    char *arena = nullptr;
    size_t arena_size = 0, used = 0;
    char *stage = nullptr;                                   // page-locked host staging (packed calls)
    size_t stage_size = 0;
    static constexpr size_t keep_limit = (size_t)1 << 30;   // larger arenas are not kept between calls

    // Arrays that stay on the device between calls of this thread, as whole-window copies keyed on the host
    // pointers and every extent:
    //  * residency cache (amt_host_cache_enable): the inputs that are constant over the acoustic sub-steps of a
    //    Runge-Kutta stage -- the five 3-D ones ww_1, u_1, v_1, t_1, ft, and (r04) the eight 2-D and four 1-D ones
    //    mut, muu, muv, mu_tend, msfuy, msfvx_inv, msftx, msfty, dnw, fnm, fnp, rdnw -- go up again only after
    //    amt_host_invalidate;
    //  * deferred outputs (amt_host_defer): ww, t, t_ave, mu, muave, muts, mudf stay on the device after the call;
    //    the device copy is the truth until amt_host_fetch brings the window's cells down (or amt_host_invalidate
    //    says the host array was rewritten and is the truth again); the in/out ones (level 1 of ww, t, mu) go up
    //    only while the device copy is not valid.
    static constexpr int NKEEP = 24;
    static constexpr int NCACHE = 17;                         // the first NCACHE entries are the cached inputs
    struct Kept {
        bool enabled = false, check = false;                  // cache on; checksum / canary debug mode
        bool defer_all = false;
        const void *defer_ptr[16] = {};
        int ndefer = 0;
        char *buf[NKEEP] = {};
        size_t bytes[NKEEP] = {};
        bool active[NKEEP] = {};                              // kept by the calls of the current key
        bool valid[NKEEP] = {};                               // the device copy is current
        bool stale[NKEEP] = {};                               // deferred: the device copy is NEWER than the host array
        bool poisoned[NKEEP] = {};                            // deferred in/out: a FAILED call may have advanced part of the device copy
        uint64_t sum[NKEEP] = {};                             // check mode: checksum of what was uploaded / of the canary
        const void *host[NKEEP] = {};
        int key[16] = {};                                     // element size, extents, window, device
        // geometry of the window the copies cover (amt_host_fetch works from it)
        size_t esize = 0;
        long idim = 0, kdim = 0, i0 = 0, ni = 0, k1 = 0, nk = 0, j_start = 0, j_end = 0, jms = 0;
        bool deferred(const void *ptr) const
        {
            if (defer_all) return true;
            for (int q = 0; q < ndefer; ++q)
                if (defer_ptr[q] == ptr) return true;
            return false;
        }
        bool any_stale() const
        {
            for (int r = 0; r < NKEEP; ++r)
                if (stale[r]) return true;
            return false;
        }
        void drop()
        {
            for (int r = 0; r < NKEEP; ++r) {
                if (buf[r]) (void)hipFree(buf[r]);
                buf[r] = nullptr; bytes[r] = 0; active[r] = valid[r] = stale[r] = poisoned[r] = false; host[r] = nullptr; sum[r] = 0;
            }
        }
    } res;

    void release()
    {
        if (device < 0) return;
        int prev = -1;
        const bool sw = hipGetDevice(&prev) == hipSuccess && prev != device && hipSetDevice(device) == hipSuccess;
        for (hipStream_t st : {up, comp, down})
            if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        for (int s = 0; s < 2; ++s)
            for (hipEvent_t e : {uploaded[s], computed[s], drained[s]})
                if (e) (void)hipEventDestroy(e);
        if (arena) (void)hipFree(arena);
        if (stage) (void)hipHostFree(stage);
        Kept keep = res;                                     // the thread's settings outlive its buffers
        res.drop();
        if (sw) (void)hipSetDevice(prev);
        *this = HostWorkspace();
        res.enabled = keep.enabled; res.check = keep.check; res.defer_all = keep.defer_all; res.ndefer = keep.ndefer;
        memcpy(res.defer_ptr, keep.defer_ptr, sizeof res.defer_ptr);
    }
    // A worker thread's workspace is freed when the thread ends.  The main thread's destructor
    // runs at process exit only, possibly from a signal path with a HIP call on the stack: leave
    // that one to the operating system.
    ~HostWorkspace()
    {
        if ((long)syscall(SYS_gettid) != (long)getpid()) release();
    }

    hipError_t prepare(int dev, size_t bytes)
    {
        if (device != dev) {
            release();
            device = dev;
            hipError_t e = hipSuccess;
            for (hipStream_t *st : {&up, &comp, &down})
                if (e == hipSuccess) e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
            for (int s = 0; s < 2; ++s)
                for (hipEvent_t *ev : {&uploaded[s], &computed[s], &drained[s]})
                    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
            if (e != hipSuccess) { release(); return e; }
        }
        if (bytes > arena_size) {
            if (arena) (void)hipFree(arena);
            arena = nullptr;
            arena_size = 0;
            hipError_t e = hipMalloc((void **)&arena, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); arena = nullptr; return e; }
            arena_size = bytes;
        }
        used = 0;
        return hipSuccess;
    }
    void *take(size_t bytes)
    {
        const size_t off = (used + 255) & ~(size_t)255;
        used = off + bytes;
        return arena + off;
    }
    hipError_t stage_reserve(size_t bytes)
    {
        if (bytes <= stage_size) return hipSuccess;
        if (stage) (void)hipHostFree(stage);
        stage = nullptr;
        stage_size = 0;
        hipError_t e = hipHostMalloc((void **)&stage, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); stage = nullptr; return e; }
        stage_size = bytes;
        return hipSuccess;
    }
    // end of a call: nothing may still be in flight towards the caller's arrays
    void finish()
    {
        for (hipStream_t st : {up, comp, down})
            if (st) (void)hipStreamSynchronize(st);
        if (arena_size > keep_limit) {
            (void)hipFree(arena);
            arena = nullptr;
            arena_size = 0;
        }
    }
};
thread_local HostWorkspace tl_workspace;
thread_local long tl_oneshot_calls = 0;                       // one-shot calls of this thread, either precision (test hook below)
struct WorkspaceScope {
    HostWorkspace &ws;
    ~WorkspaceScope() { ws.finish(); }
};
}  // namespace


// item index (argument position among the 26 arrays) and rank of every kept field: first the cached inputs, then the
// deferrable outputs
static const int kKeepField[HostWorkspace::NKEEP] = {1, 3, 5, 14, 16, 7, 10, 11, 17, 22, 23, 24, 25, 18, 19, 20, 21,
                                                     /* deferred: */ 0, 13, 15, 6, 8, 9, 12};
static const int kKeepRank[HostWorkspace::NKEEP] = {3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 3, 3, 3, 2, 2, 2, 2};
static const char *const kKeepName[HostWorkspace::NKEEP] = {"ww_1", "u_1", "v_1", "t_1", "ft", "mut", "muu", "muv", "mu_tend", "msfuy",
                                                            "msfvx_inv", "msftx", "msfty", "dnw", "fnm", "fnp", "rdnw",
                                                            "ww", "t", "t_ave", "mu", "muave", "muts", "mudf"};

// 64-bit checksum of a byte range (check mode only: it reads the whole range; byte-wise loads of the head and tail,
// memcpy-based word loads in between: the range need not be 8-byte aligned)
static uint64_t amt_host_sum(const void *p, size_t bytes)
{
    const unsigned char *q = static_cast<const unsigned char *>(p);
    uint64_t a = 0x9e3779b97f4a7c15ull, b = 0;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        memcpy(&w, q + i, 8);
        a = (a ^ w) * 0x100000001b3ull;
        b += w + (a >> 29);
    }
    for (; i < bytes; ++i) a = (a ^ q[i]) * 0x100000001b3ull;
    return a ^ (b << 1);
}

namespace {
// The window's cells of kept field r as a list of (host address, device address, bytes) runs is never built: the
// strided copies below walk it.  dir > 0: device -> host (fetch); the canary and the checksum walk the host side.
const uint64_t kCanary64 = 0x7ff8dead0badc0deull;             // a quiet NaN in fp64
const uint32_t kCanary32 = 0x7fc0dead;                        // a quiet NaN in fp32

// bring the window's cells of kept field r down to its host array (the strided copies of a normal call's download)
hipError_t amt_keep_fetch(HostWorkspace &ws, int r)
{
    HostWorkspace::Kept &k = ws.res;
    if (!k.buf[r] || !k.stale[r]) return hipSuccess;
    char *host = static_cast<char *>(const_cast<void *>(k.host[r]));
    const size_t es = k.esize, idim = (size_t)k.idim, kdim = (size_t)k.kdim;
    const size_t nj = (size_t)(k.j_end - k.j_start + 1);
    hipError_t e;
    if (kKeepRank[r] == 3) {
        if (k.nk <= 0) { k.stale[r] = false; return hipSuccess; }
        hipMemcpy3DParms cp;
        memset(&cp, 0, sizeof cp);
        cp.srcPtr = make_hipPitchedPtr(k.buf[r], idim * es, idim, kdim);
        cp.dstPtr = make_hipPitchedPtr(host, idim * es, idim, kdim);
        cp.srcPos = make_hipPos((size_t)k.i0 * es, (size_t)k.k1, 1);
        cp.dstPos = make_hipPos((size_t)k.i0 * es, (size_t)k.k1, (size_t)(k.j_start - k.jms));
        cp.extent = make_hipExtent((size_t)k.ni * es, (size_t)k.nk, nj);
        cp.kind = hipMemcpyDeviceToHost;
        e = hipMemcpy3DAsync(&cp, ws.down);
    } else {
        e = hipMemcpy2DAsync(host + ((size_t)(k.j_start - k.jms) * idim + (size_t)k.i0) * es, idim * es,
                             k.buf[r] + (idim + (size_t)k.i0) * es, idim * es, (size_t)k.ni * es, nj, hipMemcpyDeviceToHost, ws.down);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ws.down);
    if (e == hipSuccess) k.stale[r] = false;
    return e;
}

// check mode: the window's cells of a deferred host array are overwritten with NaN canaries after the call (a consumer
// that reads the stale host array computes NaNs instead of silently using old values) -- or, with fill = false, checked
// to still hold them (a host write without amt_host_invalidate would otherwise be lost)
bool amt_keep_canary(HostWorkspace::Kept &k, int r, bool fill)
{
    char *host = static_cast<char *>(const_cast<void *>(k.host[r]));
    const size_t es = k.esize, idim = (size_t)k.idim, kdim = (size_t)k.kdim;
    const size_t levels = kKeepRank[r] == 3 ? (size_t)k.nk : 1, kd = kKeepRank[r] == 3 ? kdim : 1, k1 = kKeepRank[r] == 3 ? (size_t)k.k1 : 0;
    for (long j = k.j_start; j <= k.j_end; ++j)
        for (size_t l = 0; l < levels; ++l) {
            char *row = host + ((((size_t)(j - k.jms)) * kd + k1 + l) * idim + (size_t)k.i0) * es;
            for (long i = 0; i < k.ni; ++i) {
                if (fill) {
                    if (es == 8) memcpy(row + (size_t)i * 8, &kCanary64, 8);
                    else memcpy(row + (size_t)i * 4, &kCanary32, 4);
                } else if (memcmp(row + (size_t)i * es, es == 8 ? (const void *)&kCanary64 : (const void *)&kCanary32, es) != 0) {
                    return false;
                }
            }
        }
    return true;
}

// everything the device holds that the host does not: brought down before the copies are given up
int amt_keep_flush(HostWorkspace &ws)
{
    if (ws.device < 0 || !ws.res.any_stale()) return AMT_OK;
    DeviceScope dev(ws.device);
    for (int r = 0; r < HostWorkspace::NKEEP; ++r) AMT_HIP(amt_keep_fetch(ws, r));
    return AMT_OK;
}
}  // namespace

static int host_release_local()
{
    const int rc = amt_keep_flush(tl_workspace);              // what only the device holds comes down before the buffers go
    tl_workspace.release();
    return rc;
}

static int host_cache_enable_local(int on)
{
    HostWorkspace &ws = tl_workspace;
    if (!on && ws.res.enabled && ws.device >= 0) {
        DeviceScope dev(ws.device);
        for (hipStream_t st : {ws.up, ws.comp, ws.down})
            if (st) (void)hipStreamSynchronize(st);
        for (int r = 0; r < HostWorkspace::NCACHE; ++r) {        // the cached inputs go; deferred outputs are another setting
            if (ws.res.buf[r]) (void)hipFree(ws.res.buf[r]);
            ws.res.buf[r] = nullptr; ws.res.bytes[r] = 0; ws.res.active[r] = ws.res.valid[r] = false; ws.res.host[r] = nullptr;
        }
    }
    ws.res.enabled = on != 0;
    return AMT_OK;
}

static int host_cache_check_local(int on)
{
    tl_workspace.res.check = on != 0;
    return AMT_OK;
}

static int host_invalidate_local(const void *host_ptr)
{
    // "the host array was rewritten": a cached input goes up again with the next call; for a deferred output the HOST
    // is the truth again (whatever the device still held for it is given up)
    HostWorkspace::Kept &r = tl_workspace.res;
    for (int q = 0; q < HostWorkspace::NKEEP; ++q)
        if (!host_ptr || r.host[q] == host_ptr) { r.valid[q] = false; r.stale[q] = false; r.poisoned[q] = false; }
    return AMT_OK;                                            // an array that is not kept is uploaded anyway
}

static int host_defer_local(const void *host_ptr, int on)
{
    HostWorkspace &ws = tl_workspace;
    HostWorkspace::Kept &k = ws.res;
    if (on) {
        if (!host_ptr) { k.defer_all = true; return AMT_OK; }
        if (k.deferred(host_ptr)) return AMT_OK;
        if (k.ndefer >= 16) return amt_fail(AMT_ERR_INVALID_ARG, "amt_host_defer: more than 16 deferred arrays");
        k.defer_ptr[k.ndefer++] = host_ptr;
        return AMT_OK;
    }
    // off: what the device holds for it comes down first, then the array is an ordinary output again
    for (int r = HostWorkspace::NCACHE; r < HostWorkspace::NKEEP; ++r)
        if (k.buf[r] && (!host_ptr || k.host[r] == host_ptr)) {
            if (k.stale[r] && ws.device >= 0) {
                DeviceScope dev(ws.device);
                AMT_HIP(amt_keep_fetch(ws, r));
            }
            k.valid[r] = false;                               // the next call uploads from the host again
        }
    if (!host_ptr) { k.defer_all = false; k.ndefer = 0; return AMT_OK; }
    if (k.defer_all) return amt_fail(AMT_ERR_INVALID_ARG, "amt_host_defer(ptr, 0) after amt_host_defer(NULL, 1): turn all off with NULL");
    for (int q = 0; q < k.ndefer; ++q)
        if (k.defer_ptr[q] == host_ptr) { k.defer_ptr[q] = k.defer_ptr[--k.ndefer]; break; }
    return AMT_OK;
}

static int host_fetch_local(const void *host_ptr)
{
    HostWorkspace &ws = tl_workspace;
    if (ws.device < 0) return AMT_OK;
    DeviceScope dev(ws.device);
    for (int r = HostWorkspace::NCACHE; r < HostWorkspace::NKEEP; ++r)
        if ((!host_ptr || ws.res.host[r] == host_ptr) && ws.res.poisoned[r])
            return amt_fail(AMT_ERR_PRECONDITION, "deferred output %s: its device copy is undefined since a call failed part-way (some rows may be "
                            "a sub-step ahead); nothing is brought down -- amt_host_invalidate(ptr) makes the host array the truth again", kKeepName[r]);
    for (int r = HostWorkspace::NCACHE; r < HostWorkspace::NKEEP; ++r)
        if (!host_ptr || ws.res.host[r] == host_ptr) AMT_HIP(amt_keep_fetch(ws, r));
    return AMT_OK;
}

static int host_stale_local(const void *host_ptr)
{
    const HostWorkspace::Kept &k = tl_workspace.res;
    for (int r = HostWorkspace::NCACHE; r < HostWorkspace::NKEEP; ++r)
        if (k.poisoned[r] && (!host_ptr || k.host[r] == host_ptr)) return -1;      // undefined after a failed call
    for (int r = HostWorkspace::NCACHE; r < HostWorkspace::NKEEP; ++r)
        if (k.stale[r] && (!host_ptr || k.host[r] == host_ptr)) return 1;
    return 0;
}

// One-shot call = upload, kernel, download.  Three regimes, chosen per call:
//  * streamed (3-D arrays page-locked by the caller -- amt_host_pin / hipHostRegister /
//    hipHostMalloc, once, like the reference driver's cudaHostAlloc, advance_mu_t_driver.cu:
//    97-167): the window's j rows are cut into chunks; chunk c runs
//      H2D (its rows + one halo row each side of the five arrays the stencil reads across rows)
//      -> kernel -> D2H (the window's cells of ww, t, t_ave)
//    through device buffer set c % 2, the three stages on an upload, a compute and a download
//    stream chained by events, so that both directions of the host link stay busy;
//  * streamed with a download thread (3-D arrays pageable, more than one chunk): a copy from or
//    to pageable memory blocks its host thread, so the downloads are issued by a second thread
//    (two threads drive both directions of the link at full rate, one cannot; pinning inside
//    the call costs more than it saves: 61 vs 41 ms at 512x60x512 fp64);
//  * packed (pageable and small -- WRF patch sizes): every blocking copy costs ~0.2 ms, 29 of
//    them more than everything else, so the arrays are gathered into a page-locked staging
//    buffer that mirrors the device arena and cross the link in one copy each way.
// The 2-D and 1-D arrays (1/NK of the data) go up once before the first chunk and come down once
// after the last, packed when they are pageable.  Chunks are legal because a row's outputs depend
// on other rows' INPUTS only.  The reference does one synchronous piece per call and allocates
// and frees everything around it (advance_mu_t_no_async.cu:178-306,366-423).
static bool amt_is_pinned(const void *ptr)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

extern "C" int amt_host_pin(void *ptr, size_t bytes)
{
    if (!ptr) return amt_fail(AMT_ERR_INVALID_ARG, "null pointer");
    AMT_HIP(hipHostRegister(ptr, bytes, hipHostRegisterPortable));       // page-locked for every device (amt_host_set_devices)
    return AMT_OK;
}

extern "C" int amt_host_unpin(void *ptr)
{
    if (!ptr) return amt_fail(AMT_ERR_INVALID_ARG, "null pointer");
    AMT_HIP(hipHostUnregister(ptr));
    return AMT_OK;
}

template <typename T>
static int amt_host_call(const AmtArgs<T> &h)
{
    AmtParams<T> p;
    AmtWindow w;
    bool empty = false;
    int rc = amt_build_params(h, p, w, &empty);
    if (rc != AMT_OK || empty) return rc;
    int ndev = 0;
    AMT_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return amt_fail(AMT_ERR_NO_DEVICE, "no HIP device visible");
    int device = 0;
    AMT_HIP(hipGetDevice(&device));

    const long idim = p.idim, kdim = p.kdim;
    const size_t r3 = (size_t)idim * kdim, r2 = (size_t)idim, n1 = (size_t)kdim;   // elements per j row
    const int nj = w.j_end - w.j_start + 1;
    const size_t ni = (size_t)(p.i1 - p.i0 + 1);

    // `out`: assigned by the routine; `in`: read by it (t_ave :210 and muave, muts, mudf :152-156
    // are assigned before any use; of ww only level 1 is read, :161)
    struct Item { const T *host; int rank; bool halo; bool out; bool in; };
    enum { F_WW = 0 };
    const Item items[26] = {
        {h.ww, 3, false, true, true}, {h.ww_1, 3, false, false, true}, {h.u, 3, false, false, true},
        {h.u_1, 3, false, false, true}, {h.v, 3, true, false, true}, {h.v_1, 3, true, false, true},
        {h.mu, 2, false, true, true}, {h.mut, 2, false, false, true}, {h.muave, 2, false, true, false},
        {h.muts, 2, false, true, false}, {h.muu, 2, false, false, true}, {h.muv, 2, true, false, true},
        {h.mudf, 2, false, true, false}, {h.t, 3, false, true, true}, {h.t_1, 3, true, false, true},
        {h.t_ave, 3, false, true, false}, {h.ft, 3, false, false, true}, {h.mu_tend, 2, false, false, true},
        {h.dnw, 1, false, false, true}, {h.fnm, 1, false, false, true}, {h.fnp, 1, false, false, true},
        {h.rdnw, 1, false, false, true}, {h.msfuy, 2, false, false, true}, {h.msfvx_inv, 2, true, false, true},
        {h.msftx, 2, false, false, true}, {h.msfty, 2, false, false, true},
    };
    bool pinned = true, pinned_small = true;                  // the 3-D arrays / the 2-D and 1-D ones
    for (const Item &it : items) (it.rank == 3 ? pinned : pinned_small) &= amt_is_pinned(it.host);

    // 3-D arrays that are kept on the device between calls (cached inputs, deferred outputs: below) need no chunk buffers
    bool keep_want[26] = {};
    int nbig = 10;
    for (int r = 0; r < HostWorkspace::NKEEP; ++r) {
        const int f = kKeepField[r];
        keep_want[f] = r < HostWorkspace::NCACHE ? tl_workspace.res.enabled : tl_workspace.res.deferred(items[f].host);
        if (keep_want[f] && items[f].rank == 3) --nbig;
    }
    // chunking: ~320 MB of 3-D arrays per chunk (measured best at 1024x60x1024 fp64), counting the arrays that stream
    const char *env_rows = getenv("AMT_STREAM_ROWS");         // test/tuning knob: rows per chunk
    const bool may_thread = !pinned && amt_env_flag("AMT_STREAM_THREAD", 1);
    long rows = env_rows ? atol(env_rows) : (pinned || may_thread) ? (long)((320u << 20) / (r3 * sizeof(T) * (size_t)(nbig > 0 ? nbig : 1)) + 1) : (long)nj;
    if (rows < 1) rows = 1;
    if (rows > nj) rows = nj;
    const int nchunk = (int)((nj + rows - 1) / rows);
    const int nset = nchunk > 1 ? 2 : 1;
    const size_t crow = (size_t)rows + 2;                     // device rows per 3-D buffer set
    const size_t wrow = (size_t)nj + 2;                       // device rows of a 2-D array: the window's +-1
    bool threaded = may_thread && nchunk > 1;

    // packing (see above): the small arrays when they are pageable, the 3-D ones too when they
    // are pageable, one chunk and small
    const size_t small_bytes = 12 * (r2 * wrow * sizeof(T) + 256) + 4 * (n1 * sizeof(T) + 256);
    const size_t big_bytes = (size_t)nset * nbig * (r3 * crow * sizeof(T) + 256);
    const bool allow_pack = amt_env_flag("AMT_STREAM_PACK", 1) != 0;
    const bool pack_small = allow_pack && !pinned_small && small_bytes <= ((size_t)32 << 20);
    const bool pack_big = pack_small && !pinned && nchunk == 1 && big_bytes <= ((size_t)64 << 20);

    const bool trace = getenv("AMT_STREAM_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();

    // arena layout: [3-D inputs][3-D outputs][small outputs][small inputs] -- what comes down is
    // one contiguous range, and so is everything a packed call sends up
    HostWorkspace &ws = tl_workspace;
    // Test hook (tests/test_gpu_20_host_cache.py, the one-GPU twin of the two-device test): AMT_TEST_PRETEND_DEVICE_CHANGE=n makes
    // the n-th one-shot call of a thread behave as if the thread had moved to another device since its last call -- the workspace
    // of the "old" device is given up exactly as prepare() gives it up, only the device number stays the same.
    const long calls_of_this_thread = ++tl_oneshot_calls;
    const char *pretend_env = getenv("AMT_TEST_PRETEND_DEVICE_CHANGE");
    const bool pretend_move = pretend_env && *pretend_env && atol(pretend_env) == calls_of_this_thread && ws.device >= 0;
    if (ws.device >= 0 && (ws.device != device || pretend_move) && ws.res.any_stale()) {
        // the thread moved to another device: prepare() gives the old device's workspace up, and deferred outputs whose only
        // current copy lives there must come down to their host arrays first (nothing the device alone holds is ever dropped,
        // ADVICE r04) -- on the OLD device, whose streams still exist
        rc = amt_keep_flush(ws);
        if (rc != AMT_OK) return rc;
    }
    if (pretend_move) ws.release();                           // what prepare() does when device != ws.device
    {
        const hipError_t e = ws.prepare(device, big_bytes + small_bytes);
        if (e == hipErrorOutOfMemory) return amt_fail(AMT_ERR_ALLOC, "hipMalloc of %zu bytes failed", big_bytes + small_bytes);
        AMT_HIP(e);
    }
    WorkspaceScope scope{ws};                                 // drains the three streams on every way out
    T *dev[2][26];
    size_t out3_begin = 0, small_begin = 0, small_out_end = 0;
    for (int pass = 0; pass < 4; ++pass) {
        if (pass == 1) out3_begin = (ws.used + 255) & ~(size_t)255;
        if (pass == 2) small_begin = (ws.used + 255) & ~(size_t)255;
        if (pass == 3) small_out_end = (ws.used + 255) & ~(size_t)255;
        for (int f = 0; f < 26; ++f) {
            const Item &it = items[f];
            const bool big = it.rank == 3;
            if (big != (pass < 2) || it.out != (pass == 1 || pass == 2)) continue;
            if (big && keep_want[f]) {
                dev[0][f] = dev[1][f] = nullptr;              // lives in its kept whole-window copy
            } else if (big) {
                for (int s = 0; s < nset; ++s) dev[s][f] = static_cast<T *>(ws.take(r3 * crow * sizeof(T)));
                if (nset == 1) dev[1][f] = dev[0][f];
            } else {
                dev[0][f] = dev[1][f] = static_cast<T *>(ws.take((it.rank == 2 ? r2 * wrow : n1) * sizeof(T)));
            }
        }
    }
    const size_t arena_end = ws.used;

    // ---- arrays kept on the device between calls: cached inputs and deferred outputs (whole-window copies) -----
    int res_of[26];
    for (int f = 0; f < 26; ++f) res_of[f] = -1;
    HostWorkspace::Kept &res = ws.res;
    bool res_upload[HostWorkspace::NKEEP] = {};
    long res_hi[HostWorkspace::NKEEP];                         // highest row uploaded so far in this call
    for (int r = 0; r < HostWorkspace::NKEEP; ++r) res_hi[r] = (long)w.j_start - 2;
    {
        bool want[HostWorkspace::NKEEP];
        bool any_want = false;
        for (int r = 0; r < HostWorkspace::NKEEP; ++r) {
            want[r] = keep_want[kKeepField[r]];
            any_want = any_want || want[r];
        }
        const int key[16] = {(int)sizeof(T), h.ims, h.ime, h.kms, h.kme, h.jms, h.jme, w.j_start, w.j_end, p.i0, p.i1, p.nk, p.k1, device, 0, 0};
        bool same = memcmp(res.key, key, sizeof key) == 0;
        for (int r = 0; r < HostWorkspace::NKEEP && same; ++r)
            same = want[r] == res.active[r] && (!want[r] || res.host[r] == items[kKeepField[r]].host);
        if (!same) {
            // another patch, another layout, other arrays or other settings: what only the device holds goes down to the
            // arrays it belongs to, then start over (nothing is in flight between calls)
            rc = amt_keep_flush(ws);
            if (rc != AMT_OK) return rc;
            res.drop();
            memcpy(res.key, key, sizeof key);
            res.esize = sizeof(T); res.idim = idim; res.kdim = kdim; res.i0 = p.i0; res.ni = (long)ni; res.k1 = p.k1; res.nk = p.nk;
            res.j_start = w.j_start; res.j_end = w.j_end; res.jms = h.jms;
            for (int r = 0; r < HostWorkspace::NKEEP; ++r) {
                if (!want[r]) continue;
                const size_t each = (kKeepRank[r] == 3 ? r3 * wrow : kKeepRank[r] == 2 ? r2 * wrow : n1) * sizeof(T);
                const hipError_t e = hipMalloc((void **)&res.buf[r], each);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    res.drop();
                    memset(res.key, 0, sizeof res.key);
                    return amt_fail(AMT_ERR_ALLOC, "kept arrays: hipMalloc of %zu bytes for %s failed (amt_host_cache_enable(0) / "
                                    "amt_host_defer(NULL, 0) run without them)", each, kKeepName[r]);
                }
                res.bytes[r] = each;
                res.active[r] = true;
                res.host[r] = items[kKeepField[r]].host;
            }
        }
        for (int r = 0; r < HostWorkspace::NKEEP && any_want; ++r) {
            if (!res.active[r]) continue;
            const int f = kKeepField[r];
            res_of[f] = r;
            if (res.poisoned[r])
                return amt_fail(AMT_ERR_PRECONDITION, "deferred output %s: an earlier call failed part-way and its device copy is undefined; "
                                "call amt_host_invalidate for it (the host array, as last fetched, is the truth again) before the next sub-step",
                                kKeepName[r]);
            res_upload[r] = !res.valid[r] && items[f].in;
            if (!res.check) continue;
            if (r < HostWorkspace::NCACHE) {
                // debug mode: a cached array that changed on the host without amt_host_invalidate is an error.  The
                // checksum covers exactly the rows a call uploads (rows j_start-1 / j_end+1 only for the halo arrays)
                const int lo = items[f].halo ? w.j_start - 1 : w.j_start, hi = items[f].halo ? w.j_end + 1 : w.j_end;
                const uint64_t now_sum = kKeepRank[r] == 3 ? amt_host_sum(items[f].host + (size_t)(lo - h.jms) * r3, (size_t)(hi - lo + 1) * r3 * sizeof(T))
                                       : kKeepRank[r] == 2 ? amt_host_sum(items[f].host + (size_t)(w.j_start - 1 - h.jms) * r2, r2 * wrow * sizeof(T))
                                                           : amt_host_sum(items[f].host, n1 * sizeof(T));
                if (res.valid[r] && now_sum != res.sum[r]) {
                    res.valid[r] = false;
                    return amt_fail(AMT_ERR_PRECONDITION, "residency cache: %s changed on the host since it was uploaded "
                                    "but amt_host_invalidate was not called for it (amt_host_cache_check)", kKeepName[r]);
                }
                res.sum[r] = now_sum;
            } else if (res.stale[r] && !amt_keep_canary(res, r, false)) {
                // debug mode: the host array of a deferred output was written while the device held the truth
                return amt_fail(AMT_ERR_PRECONDITION, "deferred output %s was written on the host while its device copy was newer: "
                                "call amt_host_invalidate for it (the host is the truth) or amt_host_fetch before writing "
                                "(amt_host_cache_check)", kKeepName[r]);
            }
        }
    }
    // A call that fails after its first kernel went out (any of the early returns below) leaves the deferred in/out copies
    // (level 1 of ww, t, mu: advanced in place, chunk by chunk) partly a sub-step ahead: a retry would apply the sub-step twice
    // to those rows and a fetch would bring a half-advanced state down.  They are marked undefined.
    struct FailGuard {
        HostWorkspace::Kept &k;
        bool launched = false, ok = false;
        ~FailGuard()
        {
            if (!launched || ok) return;
            for (int r = HostWorkspace::NCACHE; r < HostWorkspace::NKEEP; ++r)
                if (k.active[r] && (k.stale[r] || k.valid[r])) { k.valid[r] = k.stale[r] = false; k.poisoned[r] = true; }
        }
    } fail_guard{res};
    // staging buffer: mirrors the arena from stage_base on
    const size_t stage_base = pack_big ? 0 : small_begin;
    char *stage = nullptr;
    if (pack_small) {
        const hipError_t e = ws.stage_reserve(arena_end - stage_base);
        if (e == hipErrorOutOfMemory) return amt_fail(AMT_ERR_ALLOC, "hipHostMalloc of %zu bytes failed", arena_end - stage_base);
        AMT_HIP(e);
        stage = ws.stage;
    }
    auto staged = [&](const T *devptr) -> T * {               // the staging twin of a device address
        return reinterpret_cast<T *>(stage + ((reinterpret_cast<const char *>(devptr) - ws.arena) - stage_base));
    };
    const double t_alloc = now();
    const hipStream_t up = ws.up, comp = ws.comp, down = ws.down;

    // ---- the 1-D and 2-D arrays, once -----------------------------------------------------------
    for (int f = 0; f < 26; ++f) {
        const Item &it = items[f];
        if (it.rank == 3 || !it.in) continue;                 // muave, muts, mudf: outputs only
        const size_t n = it.rank == 1 ? n1 : r2 * wrow;
        const T *src = it.rank == 1 ? it.host : it.host + (size_t)(w.j_start - 1 - h.jms) * r2;
        if (res_of[f] >= 0) {                                 // cached 2-D / 1-D input, deferred mu: up only while the device copy is not valid
            if (res_upload[res_of[f]])
                AMT_HIP(hipMemcpyAsync(res.buf[res_of[f]], src, n * sizeof(T), hipMemcpyHostToDevice, up));
            continue;
        }
        if (pack_small) memcpy(staged(dev[0][f]), src, n * sizeof(T));
        else AMT_HIP(hipMemcpyAsync(dev[0][f], src, n * sizeof(T), hipMemcpyHostToDevice, up));
    }
    if (pack_small && !pack_big)
        AMT_HIP(hipMemcpyAsync(ws.arena + small_begin, stage, arena_end - small_begin, hipMemcpyHostToDevice, up));

    // chunk c: rows c0..c1 of the window
    auto chunk_rows = [&](int c, int &c0, int &c1) {
        c0 = w.j_start + (int)(c * rows);
        c1 = (c0 + rows - 1 < w.j_end) ? (int)(c0 + rows - 1) : w.j_end;
    };
    // chunk c's window cells of ww, t, t_ave: nothing else of the host arrays is touched
    auto download = [&](int c) -> hipError_t {
        const int s = c % nset;
        int c0, c1;
        chunk_rows(c, c0, c1);
        hipError_t e = hipStreamWaitEvent(down, ws.computed[s], 0);
        for (int f = 0; f < 26 && p.nk > 0 && e == hipSuccess; ++f) {
            const Item &it = items[f];
            if (!it.out || it.rank != 3 || res_of[f] >= 0) continue;   // deferred outputs stay on the device
            hipMemcpy3DParms cp;
            memset(&cp, 0, sizeof cp);
            cp.srcPtr = make_hipPitchedPtr(dev[s][f], (size_t)idim * sizeof(T), (size_t)idim, (size_t)kdim);
            cp.dstPtr = make_hipPitchedPtr(const_cast<T *>(it.host), (size_t)idim * sizeof(T), (size_t)idim, (size_t)kdim);
            cp.srcPos = make_hipPos((size_t)p.i0 * sizeof(T), (size_t)p.k1, 1);
            cp.dstPos = make_hipPos((size_t)p.i0 * sizeof(T), (size_t)p.k1, (size_t)(c0 - h.jms));
            cp.extent = make_hipExtent(ni * sizeof(T), (size_t)p.nk, (size_t)(c1 - c0 + 1));
            cp.kind = hipMemcpyDeviceToHost;
            e = hipMemcpy3DAsync(&cp, down);
        }
        return e != hipSuccess ? e : hipEventRecord(ws.drained[s], down);
    };

    // download thread: takes chunks in order as the main thread reports them launched, reports
    // them drained so that their buffer set can be reused
    struct Downloader {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        int launched = 0, drained = 0;
        bool stop = false;
        hipError_t err = hipSuccess;
        ~Downloader()
        {
            { std::lock_guard<std::mutex> lk(m); stop = true; }
            cv.notify_all();
            if (th.joinable()) th.join();
        }
    } dl;
    if (threaded) try {
        dl.th = std::thread([&, device] {
            hipError_t e = hipSetDevice(device);
            for (int c = 0; c < nchunk; ++c) {
                {
                    std::unique_lock<std::mutex> lk(dl.m);
                    dl.cv.wait(lk, [&] { return dl.stop || dl.launched > c; });
                    if (dl.launched <= c) return;
                }
                if (e == hipSuccess) e = download(c);
                if (e == hipSuccess) e = hipStreamSynchronize(down);
                {
                    std::lock_guard<std::mutex> lk(dl.m);
                    if (e != hipSuccess) dl.err = e;
                    dl.drained = c + 1;
                }
                dl.cv.notify_all();
            }
        });
    } catch (const std::exception &) {
        threaded = false;                                     // no thread to be had: the downloads are queued from this one (the pageable order)
    }

    // ---- the chunks -----------------------------------------------------------------------------
    for (int c = 0; c < nchunk; ++c) {
        const int s = c % nset;
        int c0, c1;
        chunk_rows(c, c0, c1);
        const int ja = c0 - 1;                                // device row 0 <-> Fortran row ja
        if (c >= nset) {                                      // set s is free again
            if (threaded) {
                std::unique_lock<std::mutex> lk(dl.m);
                dl.cv.wait(lk, [&] { return dl.drained > c - nset; });
                if (dl.err != hipSuccess) break;
            } else {
                AMT_HIP(hipStreamWaitEvent(up, ws.drained[s], 0));
            }
        }
        for (int f = 0; f < 26; ++f) {
            const Item &it = items[f];
            if (it.rank != 3 || !it.in) continue;
            if (f == F_WW) {                                  // level 1 only
                const T *src = it.host + (size_t)(c0 - h.jms) * r3 + (size_t)p.k1 * idim;
                if (res_of[f] >= 0) {                             // deferred: goes up only while the device copy is not valid
                    if (res_upload[res_of[f]])
                        AMT_HIP(hipMemcpy2DAsync(reinterpret_cast<T *>(res.buf[res_of[f]]) + (size_t)(c0 - (w.j_start - 1)) * r3 + (size_t)p.k1 * idim,
                                                 r3 * sizeof(T), src, r3 * sizeof(T), (size_t)idim * sizeof(T), (size_t)(c1 - c0 + 1),
                                                 hipMemcpyHostToDevice, up));
                    continue;
                }
                T *dst = dev[s][f] + r3 + (size_t)p.k1 * idim;
                if (pack_big)
                    for (int j = c0; j <= c1; ++j)
                        memcpy(staged(dst) + (size_t)(j - c0) * r3, src + (size_t)(j - c0) * r3, (size_t)idim * sizeof(T));
                else
                    AMT_HIP(hipMemcpy2DAsync(dst, r3 * sizeof(T), src, r3 * sizeof(T), (size_t)idim * sizeof(T),
                                             (size_t)(c1 - c0 + 1), hipMemcpyHostToDevice, up));
                continue;
            }
            int lo = it.halo ? c0 - 1 : c0;
            const int hi = it.halo ? c1 + 1 : c1;
            if (res_of[f] >= 0) {
                // kept: its rows go up only while the device copy is not valid, and every row ONCE per call -- a later chunk
                // starts behind what the earlier ones uploaded (its lower halo rows are already there, and the kernel of the
                // chunk before may be reading them: ADVICE r03)
                const int r = res_of[f];
                if (res_upload[r]) {
                    if (lo <= res_hi[r]) lo = (int)res_hi[r] + 1;
                    if (lo <= hi)
                        AMT_HIP(hipMemcpyAsync(reinterpret_cast<T *>(res.buf[r]) + (size_t)(lo - (w.j_start - 1)) * r3,
                                               it.host + (size_t)(lo - h.jms) * r3, (size_t)(hi - lo + 1) * r3 * sizeof(T),
                                               hipMemcpyHostToDevice, up));
                    res_hi[r] = hi;
                }
                continue;
            }
            const size_t n = (size_t)(hi - lo + 1) * r3 * sizeof(T);
            if (pack_big) memcpy(staged(dev[s][f]) + (size_t)(lo - ja) * r3, it.host + (size_t)(lo - h.jms) * r3, n);
            else AMT_HIP(hipMemcpyAsync(dev[s][f] + (size_t)(lo - ja) * r3, it.host + (size_t)(lo - h.jms) * r3, n,
                                        hipMemcpyHostToDevice, up));
        }
        if (pack_big)                                         // one chunk: everything, small arrays included
            AMT_HIP(hipMemcpyAsync(ws.arena, stage, arena_end, hipMemcpyHostToDevice, up));
        AMT_HIP(hipEventRecord(ws.uploaded[s], up));
        AmtArgs<T> d = h;
        T *q[26];
        for (int f = 0; f < 26; ++f)                          // every array as if it began at row ja
            q[f] = items[f].rank == 2 ? (res_of[f] >= 0 ? reinterpret_cast<T *>(res.buf[res_of[f]]) : dev[0][f]) + (size_t)(ja - (w.j_start - 1)) * r2
                 : items[f].rank == 1 ? (res_of[f] >= 0 ? reinterpret_cast<T *>(res.buf[res_of[f]]) : dev[0][f])
                 : res_of[f] >= 0 ? reinterpret_cast<T *>(res.buf[res_of[f]]) + (size_t)(ja - (w.j_start - 1)) * r3 : dev[s][f];
        d.ww = q[0]; d.ww_1 = q[1]; d.u = q[2]; d.u_1 = q[3]; d.v = q[4]; d.v_1 = q[5]; d.mu = q[6];
        d.mut = q[7]; d.muave = q[8]; d.muts = q[9]; d.muu = q[10]; d.muv = q[11]; d.mudf = q[12];
        d.t = q[13]; d.t_1 = q[14]; d.t_ave = q[15]; d.ft = q[16]; d.mu_tend = q[17];
        d.dnw = q[18]; d.fnm = q[19]; d.fnp = q[20]; d.rdnw = q[21]; d.msfuy = q[22];
        d.msfvx_inv = q[23]; d.msftx = q[24]; d.msfty = q[25];
        d.jms = ja; d.jme = c1 + 1; d.jts = c0; d.jte = c1;  // a tile of the same domain (global jds, jde)
        AMT_HIP(hipStreamWaitEvent(comp, ws.uploaded[s], 0));
        fail_guard.launched = true;
        rc = amt_device_call<T>(comp, AMT_VARIANT_AUTO, d);
        if (rc == AMT_OK && getenv("AMT_TEST_FAIL_AFTER_LAUNCH"))   // test hook: a failure after the first kernel went out
            rc = amt_fail(AMT_ERR_HIP, "AMT_TEST_FAIL_AFTER_LAUNCH: simulated failure after chunk %d was launched", c);
        if (rc != AMT_OK) break;
        AMT_HIP(hipEventRecord(ws.computed[s], comp));
        if (threaded) {
            { std::lock_guard<std::mutex> lk(dl.m); dl.launched = c + 1; }
            dl.cv.notify_all();
        } else if (!pack_big) {
            AMT_HIP(download(c));
        }
    }
    if (threaded) {                                           // all launched chunks drain, then the thread ends
        { std::lock_guard<std::mutex> lk(dl.m); dl.stop = true; }
        dl.cv.notify_all();
        dl.th.join();
        if (dl.err != hipSuccess && rc == AMT_OK)
            rc = amt_fail(AMT_ERR_HIP, "download of a chunk failed: %s", hipGetErrorString(dl.err));
    }

    // ---- the outputs that come down once, after the last kernel -----------------------------------
    if (rc == AMT_OK) {
        AMT_HIP(hipStreamWaitEvent(down, ws.computed[(nchunk - 1) % nset], 0));
        if (pack_small) {
            const size_t from = pack_big ? out3_begin : small_begin;
            AMT_HIP(hipMemcpyAsync(stage + (from - stage_base), ws.arena + from, small_out_end - from,
                                   hipMemcpyDeviceToHost, down));
            AMT_HIP(hipStreamSynchronize(down));
            for (int f = 0; f < 26; ++f) {                    // scatter: the window's cells only
                const Item &it = items[f];
                if (!it.out || res_of[f] >= 0) continue;      // deferred outputs stay on the device
                if (it.rank == 2) {
                    for (int j = w.j_start; j <= w.j_end; ++j)
                        memcpy(const_cast<T *>(it.host) + (size_t)(j - h.jms) * r2 + p.i0,
                               staged(dev[0][f]) + (size_t)(j - w.j_start + 1) * r2 + p.i0, ni * sizeof(T));
                } else if (pack_big) {
                    for (int j = w.j_start; j <= w.j_end; ++j)
                        for (int k = 0; k < p.nk; ++k) {
                            const size_t e = (size_t)(p.k1 + k) * idim + p.i0;
                            memcpy(const_cast<T *>(it.host) + (size_t)(j - h.jms) * r3 + e,
                                   staged(dev[0][f]) + (size_t)(j - w.j_start + 1) * r3 + e, ni * sizeof(T));
                        }
                }
            }
        } else {
            for (int f = 0; f < 26; ++f) {
                const Item &it = items[f];
                if (!it.out || it.rank != 2 || res_of[f] >= 0) continue;
                // the window's cells only (a strided copy): two host threads running tiles that split i
                // must not rewrite each other's columns with what they uploaded
                AMT_HIP(hipMemcpy2DAsync(const_cast<T *>(it.host) + (size_t)(w.j_start - h.jms) * r2 + p.i0, r2 * sizeof(T),
                                         dev[0][f] + r2 + p.i0, r2 * sizeof(T), ni * sizeof(T), (size_t)nj,
                                         hipMemcpyDeviceToHost, down));
            }
        }
    }
    const double t_enq = now();
    for (hipStream_t st : {up, comp, down}) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess && rc == AMT_OK)
            rc = amt_fail(AMT_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    fail_guard.ok = (rc == AMT_OK);
    for (int r = 0; r < HostWorkspace::NKEEP; ++r) {
        if (!res.active[r]) continue;
        if (r < HostWorkspace::NCACHE) {
            res.valid[r] = (rc == AMT_OK);                    // a cached input of a failed call simply goes up again
        } else if (rc == AMT_OK) {
            // the device copy is the truth from here on.  (After a call that FAILED once kernels were out, fail_guard marks the
            // deferred copies undefined: amt_host_fetch / the next call refuse them until amt_host_invalidate.)
            res.valid[r] = res.stale[r] = true;
            if (res.check) amt_keep_canary(res, r, true);
        }
    }
    if (trace)
        fprintf(stderr, "amt one-shot: %d chunk(s) of %ld rows, 3-D %s%s, small arrays %s; alloc %.2f ms, enqueue %.2f ms, drain %.2f ms\n",
                nchunk, rows, pinned ? "pinned" : pack_big ? "packed" : "pageable", threaded ? " + download thread" : "",
                pinned_small ? "pinned" : pack_small ? "packed" : "pageable", t_alloc - t_begin, t_enq - t_alloc, now() - t_enq);
    return rc;
}

// ---------------------------------------------------------------------------
// One call, several devices: the j range of the tile fanned over device slots (amt_host_set_devices / AMT_ONESHOT_DEVICES).
//
// The reference's host call IS the multi-GPU call: advance_mu_t_no_async.cu:108-162 splits j over `GPUs` devices inside one
// advance_mu_t(...), refills each device's halo rows from the HOST arrays (:135-160), launches and gathers per device
// (:329-390).  Here the calling thread owns one WORKER THREAD per device slot; a call cuts jts..jte into contiguous pieces
// (uneven where the rows do not divide) and every worker runs the ordinary one-shot call on its piece -- a tile of the same
// domain, global ids..jde unchanged, so the window rule clips each piece as it clips the whole (module_small_step_em.f90:
// 91-106), and the rows a piece reads across its edges come from the host arrays, as in the reference: no traffic between
// the devices at all.  A worker is a host thread, so everything a thread keeps between calls (streams, arena, residency cache,
// deferred outputs: HostWorkspace above) exists once per slot, on the slot's device, and the control calls (amt_host_cache_*,
// amt_host_defer / _fetch / _stale / _invalidate / _release) are forwarded to every slot.  The host link is the bound of the
// one-shot path: n devices on n links carry n times the rows.  Slots may name one device several
// times (how the tests run on a one-GPU box).
// ---------------------------------------------------------------------------
void amt_march_note_kernel(const char *name);                 // amt_kernel_march.hip: what amt_march_last_kernel reports
namespace {
struct FanWorker {
    int device = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, stop = false;
    int rc = AMT_OK;
    std::string error, kernel;

    void main()
    {
        (void)hipSetDevice(device);
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || has_job; });
                if (!has_job) return;                          // stop: the thread's workspace is released by its destructor
                f = job;
            }
            (void)hipSetDevice(device);
            const int r = f();
            {
                std::lock_guard<std::mutex> lk(m);
                rc = r;
                error = r == AMT_OK ? "" : amt_last_error();
                kernel = amt_march_last_kernel();
                has_job = false;
                done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> f)
    {
        { std::lock_guard<std::mutex> lk(m); job = std::move(f); has_job = true; done = false; }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};

struct Fan {
    std::vector<std::unique_ptr<FanWorker>> slots;
    bool env_checked = false;
    bool active() const { return !slots.empty(); }

    // every slot runs f on its own thread; the first failure (by slot) is the call's, with its text
    int all(const std::function<int(int slot)> &f, int *max_rc = nullptr)
    {
        for (size_t q = 0; q < slots.size(); ++q) slots[q]->post([f, q] { return f((int)q); });
        int first = AMT_OK, worst = 0;
        std::string text;
        for (auto &w : slots) {
            const int rc = w->wait();
            if (max_rc) {                                      // amt_host_stale: -1 (undefined) beats 1 (stale) beats 0
                if (rc < 0) worst = -1;
                else if (rc > 0 && worst == 0) worst = 1;
            } else if (rc != AMT_OK && first == AMT_OK) {
                first = rc;
                text = w->error;
            }
        }
        if (max_rc) { *max_rc = worst; return AMT_OK; }
        return first == AMT_OK ? AMT_OK : amt_fail(first, "%s", text.c_str());
    }
    int stop()
    {
        if (slots.empty()) return AMT_OK;
        const int rc = all([](int) { return host_release_local(); });     // what only a slot's device holds comes down first
        for (auto &w : slots) {
            { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
        slots.clear();
        return rc;
    }
    int start(int n, const int *ids)
    {
        int ndev = 0;
        AMT_HIP(hipGetDeviceCount(&ndev));
        for (int q = 0; q < n; ++q)
            if (ids[q] < 0 || ids[q] >= ndev) return amt_fail(AMT_ERR_INVALID_ARG, "amt_host_set_devices: device %d of %d visible", ids[q], ndev);
        int rc = stop();
        if (rc == AMT_OK) rc = host_release_local();           // what the calling thread's own device copies hold comes down: the slots read the host arrays
        if (rc != AMT_OK) return rc;
        const HostWorkspace::Kept &mine = tl_workspace.res;    // the calling thread's settings go to every slot
        try {                                                  // std::thread and the containers throw; nothing may leave through the C-ABI
            for (int q = 0; q < n; ++q) {
                std::unique_ptr<FanWorker> w(new FanWorker);
                w->device = ids[q];
                FanWorker *raw = w.get();
                slots.push_back(std::move(w));
                slots.back()->th = std::thread([raw] { raw->main(); });
            }
        } catch (const std::exception &e) {
            if (!slots.empty() && !slots.back()->th.joinable()) slots.pop_back();      // the slot whose thread did not start
            (void)stop();
            return amt_fail(AMT_ERR_ALLOC, "amt_host_set_devices: could not start %d slot threads: %s", n, e.what());
        }
        const bool enabled = mine.enabled, check = mine.check, defer_all = mine.defer_all;
        std::vector<const void *> deferred(mine.defer_ptr, mine.defer_ptr + mine.ndefer);
        return all([=](int) {
            int r = host_cache_enable_local(enabled);
            if (r == AMT_OK) r = host_cache_check_local(check);
            if (r == AMT_OK && defer_all) r = host_defer_local(nullptr, 1);
            for (const void *q : deferred)
                if (r == AMT_OK) r = host_defer_local(q, 1);
            return r;
        });
    }
    // A thread that ends takes its slots down in order (deferred outputs come down first).  The MAIN thread's destructor runs at
    // process exit, possibly after the HIP runtime's own teardown: there the workers are left to the operating system, like
    // the main thread's HostWorkspace.
    ~Fan()
    {
        if ((long)syscall(SYS_gettid) != (long)getpid()) { (void)stop(); return; }
        for (auto &w : slots) { w->th.detach(); (void)w.release(); }
    }
};
thread_local Fan tl_fan;

// AMT_ONESHOT_DEVICES="0,1,2" | "all": the device slots of every thread that has not called amt_host_set_devices (hosts that
// cannot add a call: the Fortran drop-in module is one CALL).  Read on a thread's first one-shot or control call.
int fan_from_env()
{
    Fan &fan = tl_fan;
    if (fan.env_checked) return AMT_OK;
    fan.env_checked = true;
    const char *e = getenv("AMT_ONESHOT_DEVICES");
    if (!e || !*e) return AMT_OK;
    std::vector<int> ids;
    if (!strcmp(e, "all")) {
        int ndev = 0;
        AMT_HIP(hipGetDeviceCount(&ndev));
        for (int d = 0; d < ndev; ++d) ids.push_back(d);
    } else {
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q) return amt_fail(AMT_ERR_INVALID_ARG, "AMT_ONESHOT_DEVICES must be a comma-separated list of device numbers or 'all', not '%s'", e);
            ids.push_back((int)v);
            q = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') return amt_fail(AMT_ERR_INVALID_ARG, "AMT_ONESHOT_DEVICES must be a comma-separated list of device numbers or 'all', not '%s'", e);
        }
    }
    if (ids.empty()) return AMT_OK;
    if (ids.size() == 1) {                                     // one device: no fan, the calls of this thread go there
        AMT_HIP(hipSetDevice(ids[0]));
        return AMT_OK;
    }
    return fan.start((int)ids.size(), ids.data());
}

template <typename T>
int amt_host_call_fanned(const AmtArgs<T> &h)
{
    Fan &fan = tl_fan;
    // the preconditions of the whole tile, once, on the calling thread (host arithmetic): a bad call fails before any piece runs
    {
        AmtParams<T> p;
        AmtWindow w;
        bool empty = false;
        const int rc = amt_build_params(h, p, w, &empty);
        if (rc != AMT_OK || empty) return rc;
    }
    const int n = (int)fan.slots.size();
    const long rows = (long)h.jte - h.jts + 1;
    const int rc = fan.all([&h, n, rows](int q) {
        const int lo = h.jts + (int)((rows * q) / n), hi = h.jts + (int)((rows * (q + 1)) / n) - 1;
        if (hi < lo) return (int)AMT_OK;                       // fewer rows than slots: this one has none
        AmtArgs<T> mine = h;
        mine.jts = lo;
        mine.jte = hi;
        return amt_host_call<T>(mine);
    });
    if (rc == AMT_OK && !fan.slots[0]->kernel.empty()) amt_march_note_kernel(fan.slots[0]->kernel.c_str());
    return rc;
}
}  // namespace

extern "C" int amt_host_set_devices(int n, const int *device_ids)
{
    if (n < 0 || (n > 0 && !device_ids) || n > 64) return amt_fail(AMT_ERR_INVALID_ARG, "bad device list");
    tl_fan.env_checked = true;                                 // an explicit call outranks AMT_ONESHOT_DEVICES
    if (n == 0) return tl_fan.stop();
    return tl_fan.start(n, device_ids);
}

extern "C" int amt_host_devices(int *device_ids, int cap)
{
    const Fan &fan = tl_fan;
    for (int q = 0; q < (int)fan.slots.size() && q < cap && device_ids; ++q) device_ids[q] = fan.slots[q]->device;
    return (int)fan.slots.size();
}

#define AMT_FAN_OR_LOCAL(local_call)                                              \
    do {                                                                           \
        if (int rc_ = fan_from_env()) return rc_;                                  \
        if (tl_fan.active()) {                                                     \
            const int rc_ = tl_fan.all([=](int) { return local_call; });           \
            if (rc_ != AMT_OK) return rc_;                                         \
        }                                                                          \
        return local_call;                                                         \
    } while (0)

extern "C" int amt_host_cache_enable(int on) { AMT_FAN_OR_LOCAL(host_cache_enable_local(on)); }
extern "C" int amt_host_cache_check(int on) { AMT_FAN_OR_LOCAL(host_cache_check_local(on)); }
extern "C" int amt_host_invalidate(const void *host_ptr) { AMT_FAN_OR_LOCAL(host_invalidate_local(host_ptr)); }
extern "C" int amt_host_defer(const void *host_ptr, int on) { AMT_FAN_OR_LOCAL(host_defer_local(host_ptr, on)); }
extern "C" int amt_host_fetch(const void *host_ptr) { AMT_FAN_OR_LOCAL(host_fetch_local(host_ptr)); }
extern "C" int amt_host_release(void) { AMT_FAN_OR_LOCAL(host_release_local()); }
extern "C" int amt_host_stale(const void *host_ptr)
{
    (void)fan_from_env();
    int worst = host_stale_local(host_ptr);
    if (tl_fan.active()) {
        int w = 0;
        (void)tl_fan.all([=](int) { return host_stale_local(host_ptr); }, &w);
        if (w < 0 || worst < 0) worst = -1;
        else if (w > 0) worst = 1;
    }
    return worst;
}

extern "C" int amt_advance_mu_t_f32(AMT_SIG(float))
{
    AMT_PACK_ARGS(float)
    if (int rc = fan_from_env()) return rc;
    return tl_fan.active() ? amt_host_call_fanned<float>(a) : amt_host_call<float>(a);
}
extern "C" int amt_advance_mu_t_f64(AMT_SIG(double))
{
    AMT_PACK_ARGS(double)
    if (int rc = fan_from_env()) return rc;
    return tl_fan.active() ? amt_host_call_fanned<double>(a) : amt_host_call<double>(a);
}
