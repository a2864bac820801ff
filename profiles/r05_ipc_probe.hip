// r05_ipc_probe.hip -- what two PROCESSES that share one MI355X can do for a halo exchange without RCCL
// (VERDICT r04 item 1): IPC memory handles (base + offset), a host-shared-memory mailbox both processes
// register with hipHostRegister, device-side signal / spin kernels across the two processes, peer copies
// through the opened handle while a CU-filling kernel runs, interprocess events.  Prints one line per test.
// The two ranks are fork()ed BEFORE anything touches HIP (no exec afterwards).
//   hipcc --offload-arch=gfx950 -O2 -o r05_ipc_probe r05_ipc_probe.hip && ./r05_ipc_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

struct Ctl {
    std::atomic<int> stage[2];
    hipIpcMemHandle_t h_base[2], h_inner[2], h_fine[2];
    size_t off[2];
    int inner_rc[2], fine_rc[2];
    hipIpcEventHandle_t evh[2];
    int ev_rc[2];
    char lines[2][40][200];
    int nlines[2];
};
static Ctl *ctl;
static int me, peer;

static void say(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    int n = ctl->nlines[me]++;
    if (n < 40) vsnprintf(ctl->lines[me][n], 200, fmt, ap);
    va_end(ap);
}
static bool arrive(int s, double timeout = 30.0)
{
    ctl->stage[me].store(s);
    auto t0 = std::chrono::steady_clock::now();
    while (ctl->stage[peer].load() < s) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout) return false;
        usleep(200);
    }
    return true;
}
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { say("FAIL %s: %s", #call, hipGetErrorString(e_)); return 1; } } while (0)

struct Mailbox { unsigned long long ready[2]; unsigned long long pulled[2]; unsigned int error[2]; unsigned long long pad[8]; };

__global__ void fill(unsigned long long *p, size_t n, unsigned long long seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed * 1000003ull + i;
}
__global__ void check(const unsigned long long *p, size_t n, size_t first, unsigned long long seed, unsigned int *bad)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != seed * 1000003ull + first + i) atomicAdd(bad, 1u);
}
__global__ void sig(unsigned long long *flag, unsigned long long v)
{
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void spin(const unsigned long long *flag, unsigned long long v, unsigned long long ticks, unsigned int *err)
{
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        if (wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(4);
    }
}
// pulls nseg segments (ulong2 granularity) in ONE launch: blockIdx.y = segment
typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
struct Segs { const v2u64 *src[8]; v2u64 *dst[8]; size_t n[8]; };
__global__ __launch_bounds__(256) void pull(Segs g)
{
    const v2u64 *src = g.src[blockIdx.y];
    v2u64 *dst = g.dst[blockIdx.y];
    const size_t n = g.n[blockIdx.y];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
// one workgroup per CU (150 KB of LDS), busy for `ticks`: the footprint of the interior launch
__global__ void hog(unsigned long long ticks)
{
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) lds[0] = 0;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static int run()
{
    int can = -1;
    CK(hipSetDevice(0));
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    say("hipDeviceAttributeCanUseStreamWaitValue = %d", can);

    // --- T2: mailbox in POSIX shared memory, registered by both processes
    char name[64];
    snprintf(name, sizeof name, "/amt_probe_%d", (int)getppid());
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { say("FAIL shm_open"); return 1; }
    if (ftruncate(fd, 4096) != 0) { say("FAIL ftruncate"); return 1; }
    Mailbox *mb = (Mailbox *)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mb == MAP_FAILED) { say("FAIL mmap"); return 1; }
    CK(hipHostRegister(mb, 4096, hipHostRegisterMapped | hipHostRegisterPortable));
    Mailbox *mbd = nullptr;
    CK(hipHostGetDevicePointer((void **)&mbd, mb, 0));
    say("mailbox: shm + hipHostRegister ok (host %p device %p)", (void *)mb, (void *)mbd);
    if (!arrive(1)) { say("FAIL peer missing at stage 1"); return 1; }
    if (me == 0) shm_unlink(name);

    // --- T3: IPC memory handles: base + offset, handle of an inner pointer, fine-grained memory
    const size_t N = (64u << 20) / 8;                       // 64 MiB of u64
    unsigned long long *buf = nullptr, *land = nullptr;
    unsigned int *bad = nullptr;
    CK(hipMalloc(&buf, N * 8));
    CK(hipMalloc(&land, N * 8));
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
    fill<<<1024, 256, 0, s>>>(buf, N, 7 + me);
    CK(hipStreamSynchronize(s));
    const size_t inner = (1u << 20) / 8 + 32;               // an address 1 MiB + 256 B into the allocation
    void *base = nullptr; size_t size = 0;
    CK(hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, buf + inner));
    ctl->off[me] = (char *)(buf + inner) - (char *)base;
    CK(hipIpcGetMemHandle(&ctl->h_base[me], base));
    ctl->inner_rc[me] = (int)hipIpcGetMemHandle(&ctl->h_inner[me], buf + inner);
    void *fine = nullptr;
    hipError_t ef = hipExtMallocWithFlags(&fine, 4096, hipDeviceMallocFinegrained);
    ctl->fine_rc[me] = ef == hipSuccess ? (int)hipIpcGetMemHandle(&ctl->h_fine[me], fine) : 1000 + (int)ef;
    say("hipMemGetAddressRange: base %p size %zu offset %zu; handle(inner ptr) rc %d; fine-grained alloc+handle rc %d",
        base, size, ctl->off[me], ctl->inner_rc[me], ctl->fine_rc[me]);
    if (!arrive(2)) { say("FAIL peer missing at stage 2"); return 1; }
    void *pbase = nullptr;
    hipError_t eo = hipIpcOpenMemHandle(&pbase, ctl->h_base[peer], hipIpcMemLazyEnablePeerAccess);
    if (eo != hipSuccess) { say("FAIL hipIpcOpenMemHandle(base): %s", hipGetErrorString(eo)); arrive(3); return 1; }
    const unsigned long long *psrc = (const unsigned long long *)((char *)pbase + ctl->off[peer]);
    const size_t row = (2u << 20) / 8;
    CK(hipMemcpyAsync(land, psrc, row * 8, hipMemcpyDeviceToDevice, s));
    check<<<256, 256, 0, s>>>(land, row, inner, 7 + peer, bad);
    unsigned int nbad = 1;
    CK(hipMemcpyAsync(&nbad, bad, 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    say("peer copy through hipIpcOpenMemHandle(base)+offset: %u wrong of %zu", nbad, row);
    if (ctl->inner_rc[peer] == 0) {
        void *pin = nullptr;
        hipError_t e2 = hipIpcOpenMemHandle(&pin, ctl->h_inner[peer], hipIpcMemLazyEnablePeerAccess);
        say("open handle(inner ptr): rc %d ptr %p (base mapping %p, base+offset %p)", (int)e2, pin, pbase, (void *)psrc);
        (void)hipGetLastError();
    }
    if (ctl->fine_rc[peer] == 0) {
        void *pf = nullptr;
        hipError_t e3 = hipIpcOpenMemHandle(&pf, ctl->h_fine[peer], hipIpcMemLazyEnablePeerAccess);
        say("open fine-grained handle: rc %d", (int)e3);
        (void)hipGetLastError();
    }
    if (!arrive(3)) { say("FAIL peer missing at stage 3"); return 1; }

    // --- T4: signal / spin kernels across the two processes through the mailbox: ping-pong
    const unsigned long long TICKS = 300000000ull;           // 3 s of the 100 MHz counter
    const int rounds = 200;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            sig<<<1, 1, 0, s>>>(&mbd->ready[0], (unsigned long long)r);
            spin<<<1, 1, 0, s>>>(&mbd->ready[1], (unsigned long long)r, TICKS, &mbd->error[0]);
        } else {
            spin<<<1, 1, 0, s>>>(&mbd->ready[0], (unsigned long long)r, TICKS, &mbd->error[1]);
            sig<<<1, 1, 0, s>>>(&mbd->ready[1], (unsigned long long)r);
        }
    }
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / rounds;
    say("device ping-pong through the shm mailbox: %.1f us per round trip, timeouts %u (all %d rounds enqueued up front)", us, mb->error[me], rounds);
    if (!arrive(4)) { say("FAIL peer missing at stage 4"); return 1; }

    // --- T5: the per-sweep protocol beside a CU-filling kernel of 1 ms in BOTH processes
    CK(hipFuncSetAttribute((const void *)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipEvent_t fork_ev, join_ev, e0, e1;
    CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 3; ++pass) {
        const unsigned long long hog_ticks = pass == 0 ? 0 : 100000ull;     // none, 1 ms, 1 ms
        const int sweeps = 50;
        const unsigned long long seq0 = 1000ull * (pass + 1);
        mb->error[me] = 0;
        if (!arrive(5 + pass)) { say("FAIL peer missing at stage %d", 5 + pass); return 1; }
        CK(hipEventRecord(e0, s));
        for (int k = 1; k <= sweeps; ++k) {
            const unsigned long long n = seq0 + k;
            CK(hipEventRecord(fork_ev, s));
            CK(hipStreamWaitEvent(s2, fork_ev, 0));
            if (hog_ticks) hog<<<256, 256, 150 * 1024, s>>>(hog_ticks);
            sig<<<1, 1, 0, s2>>>(&mbd->ready[me], n);
            spin<<<1, 1, 0, s2>>>(&mbd->ready[peer], n, TICKS, &mbd->error[me]);
            for (int c = 0; c < 6; ++c)
                CK(hipMemcpyAsync(land + c * row, psrc + c * row, row * 8, hipMemcpyDeviceToDevice, s2));
            sig<<<1, 1, 0, s2>>>(&mbd->pulled[me], n);
            spin<<<1, 1, 0, s2>>>(&mbd->pulled[peer], n, TICKS, &mbd->error[me]);
            CK(hipEventRecord(join_ev, s2));
            CK(hipStreamWaitEvent(s, join_ev, 0));
        }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemsetAsync(bad, 0, 4, s));
        check<<<256, 256, 0, s>>>(land, 6 * row, inner, 7 + peer, bad);
        CK(hipMemcpyAsync(&nbad, bad, 4, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        say("protocol x%d (sig+spin, 6 x 2 MiB peer copies, sig+spin)%s: %.1f us per sweep, timeouts %u, wrong %u",
            sweeps, hog_ticks ? " beside a 1 ms 256-WG 150KB-LDS kernel" : "", ms * 1e3 / sweeps, mb->error[me], nbad);
    }

    // --- T5b: what one copy costs: hipMemcpyAsync vs a pull kernel, peer (IPC mapping) vs local source
    {
        auto timed = [&](const char *what, int reps, auto &&body) -> int {
            body();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) body();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            say("%s: %.1f us each", what, ms * 1e3 / reps);
            return 0;
        };
        if (!arrive(20)) { say("FAIL peer missing at stage 20"); return 1; }
        if (me == 0) {                                  // one process at a time: clean numbers
            const size_t big = (32u << 20) / 8;
            timed("hipMemcpyAsync 2 MiB  peer->local", 20, [&] { (void)hipMemcpyAsync(land, psrc, row * 8, hipMemcpyDeviceToDevice, s); });
            timed("hipMemcpyAsync 2 MiB  local->local", 20, [&] { (void)hipMemcpyAsync(land, buf, row * 8, hipMemcpyDeviceToDevice, s); });
            timed("hipMemcpyAsync 32 MiB peer->local", 10, [&] { (void)hipMemcpyAsync(land, psrc, big * 8, hipMemcpyDeviceToDevice, s); });
            timed("hipMemcpyAsync 32 MiB local->local", 10, [&] { (void)hipMemcpyAsync(land, buf, big * 8, hipMemcpyDeviceToDevice, s); });
            timed("6 x hipMemcpyAsync 2 MiB peer->local", 10, [&] { for (int c = 0; c < 6; ++c) (void)hipMemcpyAsync(land + c * row, psrc + c * row, row * 8, hipMemcpyDeviceToDevice, s); });
            Segs g{};
            for (int c = 0; c < 6; ++c) { g.src[c] = (const v2u64 *)(psrc + c * row); g.dst[c] = (v2u64 *)(land + c * row); g.n[c] = row / 2; }
            timed("pull kernel 6 x 2 MiB peer->local, 8 WGs per segment", 20, [&] { pull<<<dim3(8, 6), 256, 0, s>>>(g); });
            timed("pull kernel 6 x 2 MiB peer->local, 32 WGs per segment", 20, [&] { pull<<<dim3(32, 6), 256, 0, s>>>(g); });
            for (int c = 0; c < 6; ++c) g.src[c] = (const v2u64 *)(buf + c * row);
            timed("pull kernel 6 x 2 MiB local->local, 8 WGs per segment", 20, [&] { pull<<<dim3(8, 6), 256, 0, s>>>(g); });
            timed("sig kernel alone", 50, [&] { sig<<<1, 1, 0, s>>>(&mbd->pad[0], 1ull); });
        }
        if (!arrive(21, 60.0)) { say("FAIL peer missing at stage 21"); return 1; }
        // the protocol again with ONE pull launch in place of the six copies
        Segs g{};
        for (int c = 0; c < 6; ++c) { g.src[c] = (const v2u64 *)(psrc + c * row); g.dst[c] = (v2u64 *)(land + c * row); g.n[c] = row / 2; }
        for (int pass = 0; pass < 2; ++pass) {
            const unsigned long long hog_ticks = pass == 0 ? 0 : 100000ull;
            const int sweeps = 50;
            const unsigned long long seq0 = 100000ull * (pass + 1);
            mb->error[me] = 0;
            CK(hipMemsetAsync(land, 0, 6 * row * 8, s));
            CK(hipStreamSynchronize(s));
            if (!arrive(22 + pass)) { say("FAIL peer missing at stage %d", 22 + pass); return 1; }
            CK(hipEventRecord(e0, s));
            for (int k = 1; k <= sweeps; ++k) {
                const unsigned long long n = seq0 + k;
                CK(hipEventRecord(fork_ev, s));
                CK(hipStreamWaitEvent(s2, fork_ev, 0));
                if (hog_ticks) hog<<<256, 256, 150 * 1024, s>>>(hog_ticks);
                sig<<<1, 1, 0, s2>>>(&mbd->ready[me], n);
                spin<<<1, 1, 0, s2>>>(&mbd->ready[peer], n, TICKS, &mbd->error[me]);
                pull<<<dim3(8, 6), 256, 0, s2>>>(g);
                sig<<<1, 1, 0, s2>>>(&mbd->pulled[me], n);
                spin<<<1, 1, 0, s2>>>(&mbd->pulled[peer], n, TICKS, &mbd->error[me]);
                CK(hipEventRecord(join_ev, s2));
                CK(hipStreamWaitEvent(s, join_ev, 0));
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemsetAsync(bad, 0, 4, s));
            check<<<256, 256, 0, s>>>(land, 6 * row, inner, 7 + peer, bad);
            CK(hipMemcpyAsync(&nbad, bad, 4, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            say("protocol x%d with one pull kernel (48 WGs)%s: %.1f us per sweep, timeouts %u, wrong %u",
                sweeps, hog_ticks ? " beside a 1 ms hog" : "", ms * 1e3 / sweeps, mb->error[me], nbad);
        }
    }

    // --- T6: interprocess events
    hipEvent_t ipc_ev = nullptr;
    hipError_t ee = hipEventCreateWithFlags(&ipc_ev, hipEventInterprocess | hipEventDisableTiming);
    ctl->ev_rc[me] = ee == hipSuccess ? (int)hipIpcGetEventHandle(&ctl->evh[me], ipc_ev) : 1000 + (int)ee;
    (void)hipGetLastError();
    if (!arrive(9)) { say("FAIL peer missing at stage 9"); return 1; }
    if (ctl->ev_rc[peer] == 0) {
        hipEvent_t pev = nullptr;
        hipError_t e4 = hipIpcOpenEventHandle(&pev, ctl->evh[peer]);
        say("interprocess event: create+handle rc %d, open peer's rc %d (%s)", ctl->ev_rc[me], (int)e4, hipGetErrorString(e4));
        (void)hipGetLastError();
    } else {
        say("interprocess event: create+handle rc %d, peer's rc %d", ctl->ev_rc[me], ctl->ev_rc[peer]);
    }
    arrive(10);
    (void)hipIpcCloseMemHandle(pbase);
    arrive(11);
    return 0;
}

int main()
{
    ctl = (Ctl *)mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset((void *)ctl, 0, sizeof(Ctl));
    pid_t pids[2];
    for (int r = 0; r < 2; ++r) {
        pids[r] = fork();
        if (pids[r] == 0) {
            me = r; peer = 1 - r;
            alarm(150);
            int rc = run();
            _exit(rc);
        }
    }
    int worst = 0;
    for (int r = 0; r < 2; ++r) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        printf("rank %d exit %d\n", r, rc);
        for (int n = 0; n < ctl->nlines[r] && n < 40; ++n) printf("  [%d] %s\n", r, ctl->lines[r][n]);
        if (rc > worst) worst = rc;
    }
    return worst;
}
