// amt_api.hip -- the C-ABI of include/amt_advance_mu_t.h on top of the gfx950 kernels.
//
// Host-side runtime of the path: bound checking and index normalisation
// (the job of advance_mu_t_no_async.cu:57-85 in the reference), the one-shot
// host-array drop-in (advance_mu_t_no_async.cu:178-423: alloc, H2D, launch, D2H,
// free), the device-resident drop-in, the resident domain handle and the
// synthetic-input fill.  There is deliberately no CPU compute path in this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
#include <string>
#include <vector>
#include <new>

#include "../../include/amt_advance_mu_t.h"
#include "../../include/amt_synth.h"
#include "amt_params.h"

template <typename T> hipError_t amt_launch_column(hipStream_t, const AmtParams<T> &);
template <typename T> hipError_t amt_launch_march(hipStream_t, const AmtParams<T> &);
template <typename T> bool amt_march_supported(const AmtParams<T> &);

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int amt_fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

#define AMT_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return amt_fail(e_ == hipErrorNoDevice ? AMT_ERR_NO_DEVICE : AMT_ERR_HIP,   \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                            __FILE__, __LINE__);                                        \
    } while (0)

extern "C" const char *amt_version(void) { return "amt-advance_mu_t 0.1 (gfx950)"; }

extern "C" const char *amt_status_string(int s)
{
    switch (s) {
    case AMT_OK: return "ok";
    case AMT_ERR_HIP: return "HIP runtime error";
    case AMT_ERR_PRECONDITION: return "bounds violate the preconditions of advance_mu_t";
    case AMT_ERR_INVALID_ARG: return "invalid argument";
    case AMT_ERR_NO_DEVICE: return "no HIP device";
    case AMT_ERR_ALLOC: return "allocation failed";
    case AMT_ERR_COMM: return "RCCL error";
    default: return "unknown status";
    }
}

extern "C" const char *amt_last_error(void) { return g_last_error.c_str(); }

extern "C" int amt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int amt_compute_window(int periodic_x, int specified, int nested,
                                  int ids, int ide, int jds, int jde,
                                  int its, int ite, int jts, int jte, int kts, int kte,
                                  int *i_start, int *i_end, int *j_start, int *j_end,
                                  int *k_start, int *k_end)
{
    const AmtWindow w = amt_window(periodic_x, specified, nested, ids, ide, jds, jde,
                                   its, ite, jts, jte, kts, kte);
    if (i_start) *i_start = w.i_start;
    if (i_end) *i_end = w.i_end;
    if (j_start) *j_start = w.j_start;
    if (j_end) *j_end = w.j_end;
    if (k_start) *k_start = w.k_start;
    if (k_end) *k_end = w.k_end;
    return AMT_OK;
}

// ---------------------------------------------------------------------------
// argument bundle shared by the entry points
// ---------------------------------------------------------------------------
template <typename T>
struct AmtArgs {
    T *ww; const T *ww_1, *u, *u_1, *v, *v_1;
    T *mu; const T *mut; T *muave, *muts; const T *muu, *muv;
    T *mudf, *t; const T *t_1; T *t_ave; const T *ft, *mu_tend;
    T rdx, rdy, dts, epssm;
    const T *dnw, *fnm, *fnp, *rdnw, *msfuy, *msfvx_inv, *msftx, *msfty;
    int periodic_x, specified, nested;
    int ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte;
};

// Checks the preconditions and rebases the Fortran bounds to memory-relative
// zero-based ones (cf. advance_mu_t_no_async.cu:57-85).  *empty is set when the
// compute window holds no column (then nothing may be dereferenced).
template <typename T>
static int amt_build_params(const AmtArgs<T> &a, AmtParams<T> &p, AmtWindow &w, bool *empty)
{
    w = amt_window(a.periodic_x, a.specified, a.nested, a.ids, a.ide, a.jds, a.jde,
                   a.its, a.ite, a.jts, a.jte, a.kts, a.kte);
    *empty = (w.i_end < w.i_start) || (w.j_end < w.j_start);
    if (a.ime < a.ims || a.jme < a.jms || a.kme < a.kms)
        return amt_fail(AMT_ERR_PRECONDITION, "empty memory extents");
    if (a.kts != 1 || a.kte != a.kde)
        return amt_fail(AMT_ERR_PRECONDITION,
                        "need kts == 1 and kte == kde (got kts=%d kte=%d kde=%d): the Fortran "
                        "uses literal levels 1,2 and wdtn(kde) (module_small_step_em.f90:159,168,221)",
                        a.kts, a.kte, a.kde);
    if (a.kms > 1 || a.kme < a.kte)
        return amt_fail(AMT_ERR_PRECONDITION, "levels 1..kte=%d not inside memory kms:kme=%d:%d",
                        a.kte, a.kms, a.kme);
    if (*empty) return AMT_OK;
    if (w.i_start - 1 < a.ims || w.i_end + 1 > a.ime)
        return amt_fail(AMT_ERR_PRECONDITION,
                        "i window %d:%d plus halo not inside memory ims:ime=%d:%d",
                        w.i_start, w.i_end, a.ims, a.ime);
    if (w.j_start - 1 < a.jms || w.j_end + 1 > a.jme)
        return amt_fail(AMT_ERR_PRECONDITION,
                        "j window %d:%d plus halo not inside memory jms:jme=%d:%d",
                        w.j_start, w.j_end, a.jms, a.jme);
    const void *ptrs[] = {a.ww, a.ww_1, a.u, a.u_1, a.v, a.v_1, a.mu, a.mut, a.muave, a.muts,
                          a.muu, a.muv, a.mudf, a.t, a.t_1, a.t_ave, a.ft, a.mu_tend, a.dnw,
                          a.fnm, a.fnp, a.rdnw, a.msfuy, a.msfvx_inv, a.msftx, a.msfty};
    for (const void *q : ptrs)
        if (!q) return amt_fail(AMT_ERR_INVALID_ARG, "null array pointer");

    p.ww = a.ww; p.mu = a.mu; p.muave = a.muave; p.muts = a.muts; p.mudf = a.mudf;
    p.t = a.t; p.t_ave = a.t_ave;
    p.ww_1 = a.ww_1; p.u = a.u; p.u_1 = a.u_1; p.v = a.v; p.v_1 = a.v_1; p.mut = a.mut;
    p.muu = a.muu; p.muv = a.muv; p.t_1 = a.t_1; p.ft = a.ft; p.mu_tend = a.mu_tend;
    p.dnw = a.dnw; p.fnm = a.fnm; p.fnp = a.fnp; p.rdnw = a.rdnw; p.msfuy = a.msfuy;
    p.msfvx_inv = a.msfvx_inv; p.msftx = a.msftx; p.msfty = a.msfty;
    p.rdx = a.rdx; p.rdy = a.rdy; p.dts = a.dts; p.epssm = a.epssm;
    p.idim = a.ime - a.ims + 1;
    p.kdim = a.kme - a.kms + 1;
    p.jstride = (long)p.idim * p.kdim;
    p.i0 = w.i_start - a.ims; p.i1 = w.i_end - a.ims;
    p.j0 = w.j_start - a.jms; p.j1 = w.j_end - a.jms;
    p.k1 = 1 - a.kms;
    p.nk = w.k_end;            // levels 1..k_end (k_end may be 0)
    return AMT_OK;
}

template <typename T>
static int amt_launch(hipStream_t stream, int variant, const AmtParams<T> &p)
{
    if (variant == AMT_VARIANT_AUTO)
        variant = amt_march_supported(p) ? AMT_VARIANT_MARCH : AMT_VARIANT_COLUMN;
    hipError_t e;
    if (variant == AMT_VARIANT_COLUMN) {
        e = amt_launch_column<T>(stream, p);
    } else if (variant == AMT_VARIANT_MARCH) {
        if (!amt_march_supported(p))
            return amt_fail(AMT_ERR_INVALID_ARG, "AMT_VARIANT_MARCH does not support nk=%d", p.nk);
        e = amt_launch_march<T>(stream, p);
    } else {
        return amt_fail(AMT_ERR_INVALID_ARG, "unknown variant %d", variant);
    }
    if (e != hipSuccess)
        return amt_fail(e == hipErrorNoDevice ? AMT_ERR_NO_DEVICE : AMT_ERR_HIP,
                        "kernel launch failed: %s", hipGetErrorString(e));
    return AMT_OK;
}

template <typename T>
static int amt_device_call(void *hip_stream, int variant, const AmtArgs<T> &a)
{
    AmtParams<T> p;
    AmtWindow w;
    bool empty = false;
    int rc = amt_build_params(a, p, w, &empty);
    if (rc != AMT_OK || empty) return rc;
    return amt_launch<T>(static_cast<hipStream_t>(hip_stream), variant, p);
}

// ---------------------------------------------------------------------------
// (1) one-shot host drop-in
// ---------------------------------------------------------------------------
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
        if (sw) (void)hipSetDevice(prev);
        *this = HostWorkspace();
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
struct WorkspaceScope {
    HostWorkspace &ws;
    ~WorkspaceScope() { ws.finish(); }
};
}  // namespace

extern "C" int amt_host_release(void)
{
    tl_workspace.release();
    return AMT_OK;
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
    AMT_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
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

    // chunking: ~320 MB of 3-D input per chunk (measured best at 1024x60x1024 fp64)
    const char *env_rows = getenv("AMT_STREAM_ROWS");         // test/tuning knob: rows per chunk
    const bool may_thread = !pinned && amt_env_flag("AMT_STREAM_THREAD", 1);
    long rows = env_rows ? atol(env_rows) : (pinned || may_thread) ? (long)((320u << 20) / (r3 * sizeof(T) * 10) + 1) : (long)nj;
    if (rows < 1) rows = 1;
    if (rows > nj) rows = nj;
    const int nchunk = (int)((nj + rows - 1) / rows);
    const int nset = nchunk > 1 ? 2 : 1;
    const size_t crow = (size_t)rows + 2;                     // device rows per 3-D buffer set
    const size_t wrow = (size_t)nj + 2;                       // device rows of a 2-D array: the window's +-1
    const bool threaded = may_thread && nchunk > 1;

    // packing (see above): the small arrays when they are pageable, the 3-D ones too when they
    // are pageable, one chunk and small
    const size_t small_bytes = 12 * (r2 * wrow * sizeof(T) + 256) + 4 * (n1 * sizeof(T) + 256);
    const size_t big_bytes = (size_t)nset * 10 * (r3 * crow * sizeof(T) + 256);
    const bool allow_pack = amt_env_flag("AMT_STREAM_PACK", 1) != 0;
    const bool pack_small = allow_pack && !pinned_small && small_bytes <= ((size_t)32 << 20);
    const bool pack_big = pack_small && !pinned && nchunk == 1 && big_bytes <= ((size_t)64 << 20);

    const bool trace = getenv("AMT_STREAM_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();

    // arena layout: [3-D inputs][3-D outputs][small outputs][small inputs] -- what comes down is
    // one contiguous range, and so is everything a packed call sends up
    HostWorkspace &ws = tl_workspace;
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
            if (big) {
                for (int s = 0; s < nset; ++s) dev[s][f] = static_cast<T *>(ws.take(r3 * crow * sizeof(T)));
                if (nset == 1) dev[1][f] = dev[0][f];
            } else {
                dev[0][f] = dev[1][f] = static_cast<T *>(ws.take((it.rank == 2 ? r2 * wrow : n1) * sizeof(T)));
            }
        }
    }
    const size_t arena_end = ws.used;
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
        if (pack_small) memcpy(staged(dev[0][f]), src, n * sizeof(T));
        else AMT_HIP(hipMemcpyAsync(dev[0][f], src, n * sizeof(T), hipMemcpyHostToDevice, up));
    }
    if (!pack_small)                                          // whole-row downloads: keep the other cells' bits
        for (int f = 0; f < 26; ++f) {
            const Item &it = items[f];
            if (it.rank != 2 || it.in) continue;
            AMT_HIP(hipMemcpyAsync(dev[0][f], it.host + (size_t)(w.j_start - 1 - h.jms) * r2, r2 * wrow * sizeof(T),
                                   hipMemcpyHostToDevice, up));
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
            if (!it.out || it.rank != 3) continue;
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
    if (threaded) {
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
                T *dst = dev[s][f] + r3 + (size_t)p.k1 * idim;
                if (pack_big)
                    for (int j = c0; j <= c1; ++j)
                        memcpy(staged(dst) + (size_t)(j - c0) * r3, src + (size_t)(j - c0) * r3, (size_t)idim * sizeof(T));
                else
                    AMT_HIP(hipMemcpy2DAsync(dst, r3 * sizeof(T), src, r3 * sizeof(T), (size_t)idim * sizeof(T),
                                             (size_t)(c1 - c0 + 1), hipMemcpyHostToDevice, up));
                continue;
            }
            const int lo = it.halo ? c0 - 1 : c0, hi = it.halo ? c1 + 1 : c1;
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
            q[f] = items[f].rank == 2 ? dev[0][f] + (size_t)(ja - (w.j_start - 1)) * r2 : dev[s][f];
        d.ww = q[0]; d.ww_1 = q[1]; d.u = q[2]; d.u_1 = q[3]; d.v = q[4]; d.v_1 = q[5]; d.mu = q[6];
        d.mut = q[7]; d.muave = q[8]; d.muts = q[9]; d.muu = q[10]; d.muv = q[11]; d.mudf = q[12];
        d.t = q[13]; d.t_1 = q[14]; d.t_ave = q[15]; d.ft = q[16]; d.mu_tend = q[17];
        d.dnw = q[18]; d.fnm = q[19]; d.fnp = q[20]; d.rdnw = q[21]; d.msfuy = q[22];
        d.msfvx_inv = q[23]; d.msftx = q[24]; d.msfty = q[25];
        d.jms = ja; d.jme = c1 + 1; d.jts = c0; d.jte = c1;  // a tile of the same domain (global jds, jde)
        AMT_HIP(hipStreamWaitEvent(comp, ws.uploaded[s], 0));
        rc = amt_device_call<T>(comp, AMT_VARIANT_AUTO, d);
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
                if (!it.out) continue;
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
                if (!it.out || it.rank != 2) continue;
                AMT_HIP(hipMemcpyAsync(const_cast<T *>(it.host) + (size_t)(w.j_start - h.jms) * r2, dev[0][f] + r2,
                                       (size_t)nj * r2 * sizeof(T), hipMemcpyDeviceToHost, down));
            }
        }
    }
    const double t_enq = now();
    for (hipStream_t st : {up, comp, down}) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess && rc == AMT_OK)
            rc = amt_fail(AMT_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    if (trace)
        fprintf(stderr, "amt one-shot: %d chunk(s) of %ld rows, 3-D %s%s, small arrays %s; alloc %.2f ms, enqueue %.2f ms, drain %.2f ms\n",
                nchunk, rows, pinned ? "pinned" : pack_big ? "packed" : "pageable", threaded ? " + download thread" : "",
                pinned_small ? "pinned" : pack_small ? "packed" : "pageable", t_alloc - t_begin, t_enq - t_alloc, now() - t_enq);
    return rc;
}

#define AMT_PACK_ARGS(T)                                                                        \
    AmtArgs<T> a;                                                                               \
    a.ww = ww; a.ww_1 = ww_1; a.u = u; a.u_1 = u_1; a.v = v; a.v_1 = v_1; a.mu = mu;            \
    a.mut = mut; a.muave = muave; a.muts = muts; a.muu = muu; a.muv = muv; a.mudf = mudf;       \
    a.t = t; a.t_1 = t_1; a.t_ave = t_ave; a.ft = ft; a.mu_tend = mu_tend;                      \
    a.rdx = rdx; a.rdy = rdy; a.dts = dts; a.epssm = epssm;                                     \
    a.dnw = dnw; a.fnm = fnm; a.fnp = fnp; a.rdnw = rdnw; a.msfuy = msfuy;                      \
    a.msfvx_inv = msfvx_inv; a.msftx = msftx; a.msfty = msfty;                                  \
    a.periodic_x = periodic_x; a.specified = specified; a.nested = nested;                      \
    a.ids = ids; a.ide = ide; a.jds = jds; a.jde = jde; a.kde = kde;                            \
    a.ims = ims; a.ime = ime; a.jms = jms; a.jme = jme; a.kms = kms; a.kme = kme;               \
    a.its = its; a.ite = ite; a.jts = jts; a.jte = jte; a.kts = kts; a.kte = kte;

#define AMT_SIG(T)                                                                              \
    T *ww, const T *ww_1, const T *u, const T *u_1, const T *v, const T *v_1,                   \
    T *mu, const T *mut, T *muave, T *muts, const T *muu, const T *muv,                         \
    T *mudf, T *t, const T *t_1, T *t_ave, const T *ft, const T *mu_tend,                       \
    T rdx, T rdy, T dts, T epssm,                                                               \
    const T *dnw, const T *fnm, const T *fnp, const T *rdnw,                                    \
    const T *msfuy, const T *msfvx_inv, const T *msftx, const T *msfty,                         \
    int periodic_x, int specified, int nested,                                                  \
    int ids, int ide, int jds, int jde, int kde,                                                \
    int ims, int ime, int jms, int jme, int kms, int kme,                                       \
    int its, int ite, int jts, int jte, int kts, int kte

extern "C" int amt_advance_mu_t_f32(AMT_SIG(float))
{
    AMT_PACK_ARGS(float)
    return amt_host_call<float>(a);
}
extern "C" int amt_advance_mu_t_f64(AMT_SIG(double))
{
    AMT_PACK_ARGS(double)
    return amt_host_call<double>(a);
}
extern "C" int amt_advance_mu_t_device_f32(void *hip_stream, int variant, AMT_SIG(float))
{
    AMT_PACK_ARGS(float)
    return amt_device_call<float>(hip_stream, variant, a);
}
extern "C" int amt_advance_mu_t_device_f64(void *hip_stream, int variant, AMT_SIG(double))
{
    AMT_PACK_ARGS(double)
    return amt_device_call<double>(hip_stream, variant, a);
}

// ---------------------------------------------------------------------------
// (4) synthetic inputs
// ---------------------------------------------------------------------------
template <typename T>
__global__ void amt_synth_fill_kernel(T *dst, int field, uint64_t seed,
                                      long idim, long kdim, long jdim,
                                      long gi0, long gk0, long gj0,
                                      long gidim, long gkdim, long gjdim)
{
    const long n = idim * kdim * jdim;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const long li = e % idim;
        const long lk = (e / idim) % kdim;
        const long lj = e / (idim * kdim);
        dst[e] = (T)amt_synth_value(field, seed, gi0 + li, gk0 + lk, gj0 + lj, gidim, gkdim, gjdim);
    }
}

static int amt_synth_shape(int field, long &idim, long &kdim, long &jdim, long &gi0, long &gk0, long &gj0)
{
    if (field < 0 || field >= AMT_F_COUNT) return amt_fail(AMT_ERR_INVALID_ARG, "bad field id %d", field);
    const int rank = amt_field_rank(field);
    if (rank == 2) { kdim = 1; gk0 = 0; }
    if (rank == 1) { idim = 1; jdim = 1; gi0 = 0; gj0 = 0; }
    if (idim < 0 || kdim < 0 || jdim < 0) return amt_fail(AMT_ERR_INVALID_ARG, "negative extent");
    return AMT_OK;
}

extern "C" int amt_synth_fill_host(int field, int dtype_bytes, void *dst, uint64_t seed,
                                   long idim, long kdim, long jdim,
                                   long gi0, long gk0, long gj0,
                                   long gidim, long gkdim, long gjdim)
{
    int rc = amt_synth_shape(field, idim, kdim, jdim, gi0, gk0, gj0);
    if (rc) return rc;
    if (!dst) return amt_fail(AMT_ERR_INVALID_ARG, "null destination");
    if (dtype_bytes != 4 && dtype_bytes != 8) return amt_fail(AMT_ERR_INVALID_ARG, "dtype_bytes must be 4 or 8");
    long e = 0;
    for (long lj = 0; lj < jdim; ++lj)
        for (long lk = 0; lk < kdim; ++lk)
            for (long li = 0; li < idim; ++li, ++e) {
                const double x = amt_synth_value(field, seed, gi0 + li, gk0 + lk, gj0 + lj, gidim, gkdim, gjdim);
                if (dtype_bytes == 8) static_cast<double *>(dst)[e] = x;
                else static_cast<float *>(dst)[e] = (float)x;
            }
    return AMT_OK;
}

extern "C" int amt_synth_fill_device(void *hip_stream, int field, int dtype_bytes, void *dst,
                                     uint64_t seed, long idim, long kdim, long jdim,
                                     long gi0, long gk0, long gj0,
                                     long gidim, long gkdim, long gjdim)
{
    int rc = amt_synth_shape(field, idim, kdim, jdim, gi0, gk0, gj0);
    if (rc) return rc;
    if (!dst) return amt_fail(AMT_ERR_INVALID_ARG, "null destination");
    if (dtype_bytes != 4 && dtype_bytes != 8) return amt_fail(AMT_ERR_INVALID_ARG, "dtype_bytes must be 4 or 8");
    const long n = idim * kdim * jdim;
    if (n == 0) return AMT_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (dtype_bytes == 8)
        hipLaunchKernelGGL(amt_synth_fill_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<double *>(dst), field, seed, idim, kdim, jdim, gi0, gk0, gj0,
                           gidim, gkdim, gjdim);
    else
        hipLaunchKernelGGL(amt_synth_fill_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<float *>(dst), field, seed, idim, kdim, jdim, gi0, gk0, gj0,
                           gidim, gkdim, gjdim);
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}

// ---------------------------------------------------------------------------
// (3) resident domain handle
// ---------------------------------------------------------------------------
namespace {
// makes the domain's device current for the duration of a call and restores the caller's
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int want)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != want) switched = (hipSetDevice(want) == hipSuccess);
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};
}  // namespace

struct amt_domain {
    int dtype_bytes = 8;
    int periodic_x = 0, specified = 0, nested = 0;
    int ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte;
    double rdx = AMT_SYNTH_RDX, rdy = AMT_SYNTH_RDY, dts = AMT_SYNTH_DTS, epssm = AMT_SYNTH_EPSSM;
    int variant = AMT_VARIANT_AUTO;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *field[AMT_F_COUNT] = {};
    size_t count(int f) const
    {
        const size_t idim = ime - ims + 1, kdim = kme - kms + 1, jdim = jme - jms + 1;
        const int r = amt_field_rank(f);
        return r == 3 ? idim * kdim * jdim : r == 2 ? idim * jdim : kdim;
    }
};

extern "C" int amt_domain_destroy(amt_domain *d)
{
    if (!d) return AMT_OK;
    DeviceScope scope(d->device);
    for (void *&q : d->field)
        if (q) { (void)hipFree(q); q = nullptr; }
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
    return AMT_OK;
}

extern "C" int amt_domain_create(amt_domain **out, int dtype_bytes,
                                 int periodic_x, int specified, int nested,
                                 int ids, int ide, int jds, int jde, int kde,
                                 int ims, int ime, int jms, int jme, int kms, int kme,
                                 int its, int ite, int jts, int jte, int kts, int kte)
{
    if (!out) return amt_fail(AMT_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (dtype_bytes != 4 && dtype_bytes != 8) return amt_fail(AMT_ERR_INVALID_ARG, "dtype_bytes must be 4 or 8");
    if (ime < ims || jme < jms || kme < kms) return amt_fail(AMT_ERR_PRECONDITION, "empty memory extents");
    int ndev = 0;
    AMT_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return amt_fail(AMT_ERR_NO_DEVICE, "no HIP device visible");
    amt_domain *d = new (std::nothrow) amt_domain;
    if (!d) return amt_fail(AMT_ERR_ALLOC, "host allocation failed");
    d->dtype_bytes = dtype_bytes;
    d->periodic_x = periodic_x; d->specified = specified; d->nested = nested;
    d->ids = ids; d->ide = ide; d->jds = jds; d->jde = jde; d->kde = kde;
    d->ims = ims; d->ime = ime; d->jms = jms; d->jme = jme; d->kms = kms; d->kme = kme;
    d->its = its; d->ite = ite; d->jts = jts; d->jte = jte; d->kts = kts; d->kte = kte;
    hipError_t e = hipGetDevice(&d->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&d->ev0);
    if (e == hipSuccess) e = hipEventCreate(&d->ev1);
    for (int f = 0; f < AMT_F_COUNT && e == hipSuccess; ++f)
        e = hipMalloc(&d->field[f], d->count(f) * (size_t)dtype_bytes);
    if (e != hipSuccess) {
        amt_domain_destroy(d);
        return amt_fail(e == hipErrorOutOfMemory ? AMT_ERR_ALLOC : AMT_ERR_HIP,
                        "amt_domain_create: %s", hipGetErrorString(e));
    }
    *out = d;
    return AMT_OK;
}

extern "C" int amt_domain_set_scalars(amt_domain *d, double rdx, double rdy, double dts, double epssm)
{
    if (!d) return amt_fail(AMT_ERR_INVALID_ARG, "null domain");
    d->rdx = rdx; d->rdy = rdy; d->dts = dts; d->epssm = epssm;
    return AMT_OK;
}

extern "C" int amt_domain_set_variant(amt_domain *d, int variant)
{
    if (!d) return amt_fail(AMT_ERR_INVALID_ARG, "null domain");
    if (variant < AMT_VARIANT_AUTO || variant > AMT_VARIANT_MARCH)
        return amt_fail(AMT_ERR_INVALID_ARG, "unknown variant %d", variant);
    d->variant = variant;
    return AMT_OK;
}

extern "C" int amt_domain_upload(amt_domain *d, int field, const void *host)
{
    if (!d || !host || field < 0 || field >= AMT_F_COUNT) return amt_fail(AMT_ERR_INVALID_ARG, "bad upload argument");
    DeviceScope scope(d->device);
    AMT_HIP(hipMemcpyAsync(d->field[field], host, d->count(field) * d->dtype_bytes, hipMemcpyHostToDevice, d->stream));
    AMT_HIP(hipStreamSynchronize(d->stream));
    return AMT_OK;
}

extern "C" int amt_domain_download(amt_domain *d, int field, void *host)
{
    if (!d || !host || field < 0 || field >= AMT_F_COUNT) return amt_fail(AMT_ERR_INVALID_ARG, "bad download argument");
    DeviceScope scope(d->device);
    AMT_HIP(hipMemcpyAsync(host, d->field[field], d->count(field) * d->dtype_bytes, hipMemcpyDeviceToHost, d->stream));
    AMT_HIP(hipStreamSynchronize(d->stream));
    return AMT_OK;
}

extern "C" int amt_domain_fill_synthetic(amt_domain *d, uint64_t seed,
                                         long gi0, long gk0, long gj0,
                                         long gidim, long gkdim, long gjdim)
{
    if (!d) return amt_fail(AMT_ERR_INVALID_ARG, "null domain");
    DeviceScope scope(d->device);
    const long idim = d->ime - d->ims + 1, kdim = d->kme - d->kms + 1, jdim = d->jme - d->jms + 1;
    for (int f = 0; f < AMT_F_COUNT; ++f) {
        int rc = amt_synth_fill_device(d->stream, f, d->dtype_bytes, d->field[f], seed,
                                       idim, kdim, jdim, gi0, gk0, gj0, gidim, gkdim, gjdim);
        if (rc) return rc;
    }
    return AMT_OK;
}

template <typename T>
static void amt_domain_args(amt_domain *d, AmtArgs<T> &a)
{
    T **f = reinterpret_cast<T **>(d->field);
    a.ww = f[AMT_F_WW]; a.ww_1 = f[AMT_F_WW_1]; a.u = f[AMT_F_U]; a.u_1 = f[AMT_F_U_1];
    a.v = f[AMT_F_V]; a.v_1 = f[AMT_F_V_1]; a.mu = f[AMT_F_MU]; a.mut = f[AMT_F_MUT];
    a.muave = f[AMT_F_MUAVE]; a.muts = f[AMT_F_MUTS]; a.muu = f[AMT_F_MUU]; a.muv = f[AMT_F_MUV];
    a.mudf = f[AMT_F_MUDF]; a.t = f[AMT_F_T]; a.t_1 = f[AMT_F_T_1]; a.t_ave = f[AMT_F_T_AVE];
    a.ft = f[AMT_F_FT]; a.mu_tend = f[AMT_F_MU_TEND];
    a.rdx = (T)d->rdx; a.rdy = (T)d->rdy; a.dts = (T)d->dts; a.epssm = (T)d->epssm;
    a.dnw = f[AMT_F_DNW]; a.fnm = f[AMT_F_FNM]; a.fnp = f[AMT_F_FNP]; a.rdnw = f[AMT_F_RDNW];
    a.msfuy = f[AMT_F_MSFUY]; a.msfvx_inv = f[AMT_F_MSFVX_INV]; a.msftx = f[AMT_F_MSFTX];
    a.msfty = f[AMT_F_MSFTY];
    a.periodic_x = d->periodic_x; a.specified = d->specified; a.nested = d->nested;
    a.ids = d->ids; a.ide = d->ide; a.jds = d->jds; a.jde = d->jde; a.kde = d->kde;
    a.ims = d->ims; a.ime = d->ime; a.jms = d->jms; a.jme = d->jme; a.kms = d->kms; a.kme = d->kme;
    a.its = d->its; a.ite = d->ite; a.jts = d->jts; a.jte = d->jte; a.kts = d->kts; a.kte = d->kte;
}

template <typename T>
static int amt_domain_step_t(amt_domain *d, int n_sweeps)
{
    AmtArgs<T> a;
    amt_domain_args<T>(d, a);
    for (int s = 0; s < n_sweeps; ++s) {
        int rc = amt_device_call<T>(d->stream, d->variant, a);
        if (rc) return rc;
    }
    return AMT_OK;
}

extern "C" int amt_domain_step(amt_domain *d, int n_sweeps)
{
    if (!d || n_sweeps < 0) return amt_fail(AMT_ERR_INVALID_ARG, "bad step argument");
    DeviceScope scope(d->device);
    return d->dtype_bytes == 8 ? amt_domain_step_t<double>(d, n_sweeps) : amt_domain_step_t<float>(d, n_sweeps);
}

extern "C" int amt_domain_step_timed(amt_domain *d, int n_sweeps, float *ms_total)
{
    if (!d || n_sweeps < 0) return amt_fail(AMT_ERR_INVALID_ARG, "bad step argument");
    DeviceScope scope(d->device);
    AMT_HIP(hipEventRecord(d->ev0, d->stream));
    int rc = amt_domain_step(d, n_sweeps);
    if (rc) return rc;
    AMT_HIP(hipEventRecord(d->ev1, d->stream));
    AMT_HIP(hipEventSynchronize(d->ev1));
    float ms = 0.f;
    AMT_HIP(hipEventElapsedTime(&ms, d->ev0, d->ev1));
    if (ms_total) *ms_total = ms;
    return AMT_OK;
}

extern "C" int amt_domain_sync(amt_domain *d)
{
    if (!d) return amt_fail(AMT_ERR_INVALID_ARG, "null domain");
    DeviceScope scope(d->device);
    AMT_HIP(hipStreamSynchronize(d->stream));
    return AMT_OK;
}

extern "C" void *amt_domain_field_ptr(amt_domain *d, int field)
{
    if (!d || field < 0 || field >= AMT_F_COUNT) return nullptr;
    return d->field[field];
}

extern "C" void *amt_domain_stream(amt_domain *d) { return d ? (void *)d->stream : nullptr; }

// ---------------------------------------------------------------------------
// (6) j-slab stepping with RCCL halos -- the native twin of patch.SlabStepper, for C / Fortran
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
// under a temporary name, then renamed), the other ranks wait for the file.  A file left by an
// earlier launch must not be taken for this one's: rank 0 removes it first, and the others ignore
// files last written more than a minute before they started waiting (use a fresh path per launch,
// e.g. derived from the launcher's port, when relaunching faster than that).
extern "C" int amt_comm_rendezvous_file(const char *path, int rank, double timeout_s, void *id_out)
{
    if (!path || !*path || !id_out || rank < 0) return amt_fail(AMT_ERR_INVALID_ARG, "bad rendezvous argument");
    if (rank == 0) {
        (void)unlink(path);
        int rc = amt_comm_unique_id(id_out);
        if (rc) return rc;
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return amt_fail(AMT_ERR_COMM, "cannot write %s", tmp.c_str());
        const size_t n = fwrite(id_out, 1, AMT_UNIQUE_ID_BYTES, f);
        fclose(f);
        if (n != AMT_UNIQUE_ID_BYTES || rename(tmp.c_str(), path) != 0)
            return amt_fail(AMT_ERR_COMM, "cannot publish %s", path);
        return AMT_OK;
    }
    const time_t entered = time(nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        struct stat st;
        if (stat(path, &st) == 0 && st.st_mtime >= entered - 60) {
            if (FILE *f = fopen(path, "rb")) {
                const size_t n = fread(id_out, 1, AMT_UNIQUE_ID_BYTES, f);
                fclose(f);
                if (n == AMT_UNIQUE_ID_BYTES) return AMT_OK;
            }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            return amt_fail(AMT_ERR_COMM, "no rendezvous file %s after %.0f s", path, timeout_s);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
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
    hipError_t e = hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking);
    for (hipEvent_t *ev : {&s->inputs_final, &s->edges_done})
        if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    for (hipEvent_t *ev : {&s->t0, &s->t1})
        if (e == hipSuccess) e = hipEventCreate(ev);
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
    // per pair of ranks the order of sends matches the order of receives on the other side
    AMT_NCCL(g_rccl.GroupStart());
    if (s->below >= 0)
        for (int f : kHaloFromAbove) { void *q = row(f, d->jts, n); AMT_NCCL(g_rccl.Send(q, n, dt, s->below, s->comm, stream)); }
    if (s->above >= 0)
        for (int f : kHaloFromBelow) { void *q = row(f, d->jte, n); AMT_NCCL(g_rccl.Send(q, n, dt, s->above, s->comm, stream)); }
    if (s->above >= 0)
        for (int f : kHaloFromAbove) { void *q = row(f, d->jte + 1, n); AMT_NCCL(g_rccl.Recv(q, n, dt, s->above, s->comm, stream)); }
    if (s->below >= 0)
        for (int f : kHaloFromBelow) { void *q = row(f, d->jts - 1, n); AMT_NCCL(g_rccl.Recv(q, n, dt, s->below, s->comm, stream)); }
    AMT_NCCL(g_rccl.GroupEnd());
    return AMT_OK;
}

template <typename T>
int amt_slab_tile(amt_slab *s, hipStream_t stream, int jts, int jte)
{
    if (jte < jts) return AMT_OK;
    AmtArgs<T> a;
    amt_domain_args<T>(s->dom, a);
    a.jts = jts; a.jte = jte;
    return amt_device_call<T>(stream, s->dom->variant, a);
}

template <typename T>
int amt_slab_step_t(amt_slab *s, int n_sweeps)
{
    amt_domain *d = s->dom;
    const int jlo = d->jts, jhi = d->jte;
    const bool lo = s->below >= 0, hi = s->above >= 0;
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
            rc = amt_slab_tile<T>(s, d->stream, in_lo, in_hi);              // interior overlaps the exchange
            if (rc) return rc;
        }
        rc = amt_slab_enqueue_exchange(s, edge_stream);
        if (rc) return rc;
        if (!s->overlap) {
            rc = amt_slab_tile<T>(s, d->stream, in_lo, in_hi);
            if (rc) return rc;
        }
        if (lo) { rc = amt_slab_tile<T>(s, edge_stream, jlo, jlo < jhi ? jlo : jhi); if (rc) return rc; }
        if (hi && (jhi > jlo || !lo)) { rc = amt_slab_tile<T>(s, edge_stream, jhi, jhi); if (rc) return rc; }
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

// ---------------------------------------------------------------------------
// (7) profiling aid: streaming copy with a chosen access width
// ---------------------------------------------------------------------------
template <typename V>
__global__ void amt_calib_copy_kernel(V *dst, const V *src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) dst[e] = src[e];
}

extern "C" int amt_calib_stream_copy(void *hip_stream, void *dst, const void *src, size_t nbytes, int bytes_per_lane)
{
    if (!dst || !src) return amt_fail(AMT_ERR_INVALID_ARG, "null pointer");
    if (bytes_per_lane != 4 && bytes_per_lane != 8 && bytes_per_lane != 16)
        return amt_fail(AMT_ERR_INVALID_ARG, "bytes_per_lane must be 4, 8 or 16");
    if (nbytes % bytes_per_lane) return amt_fail(AMT_ERR_INVALID_ARG, "nbytes not a multiple of the access width");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t n = nbytes / bytes_per_lane;
    if (n == 0) return AMT_OK;
    const unsigned blocks = 256 * 8;
    if (bytes_per_lane == 4)
        hipLaunchKernelGGL(amt_calib_copy_kernel<float>, dim3(blocks), dim3(256), 0, s, (float *)dst, (const float *)src, n);
    else if (bytes_per_lane == 8)
        hipLaunchKernelGGL(amt_calib_copy_kernel<double>, dim3(blocks), dim3(256), 0, s, (double *)dst, (const double *)src, n);
    else
        hipLaunchKernelGGL(amt_calib_copy_kernel<double2>, dim3(blocks), dim3(256), 0, s, (double2 *)dst, (const double2 *)src, n);
    AMT_HIP(hipGetLastError());
    return AMT_OK;
}
