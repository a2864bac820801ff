// profiles/vmm_probe.hip -- r03 diagnostic (GPU box): what makes one PLACEMENT of the resident arrays faster
// than another?  box_probe.py showed: same virtual addresses, fresh physical pages -> 15.4 .. 16.3 ms, and
// sub-2-MiB staggers between the arrays do nothing.  This program times the bench workload through the C-ABI
// on arrays obtained in different ways, in one process:
//   mode 0  hipMalloc per array
//   mode N  virtual-memory API: one address range per array, backed by physical handles of N MiB each
// and a set of streaming kernels (the box's own ceiling): tuned copy (16 B per lane, U loads in flight, nt),
// read-only stream.
//   hipcc -O3 --offload-arch=gfx950 profiles/vmm_probe.hip -Iinclude -Lwrf-model-cuda-sample_amd -lamt_advance_mu_t \
//         -Wl,-rpath,'$ORIGIN/../wrf-model-cuda-sample_amd' -o profiles/vmm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>
#include "amt_advance_mu_t.h"
#include "amt_synth.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(v2d *dst, const v2d *src, size_t n)       // n in 16-byte units
{
    const size_t chunk = (size_t)256 * U;
    for (size_t c = (size_t)blockIdx.x * chunk; c < n; c += (size_t)gridDim.x * chunk) {
        v2d r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t e = c + (size_t)u * 256 + threadIdx.x;
            if (e < n) r[u] = NT ? __builtin_nontemporal_load(src + e) : src[e];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t e = c + (size_t)u * 256 + threadIdx.x;
            if (e < n) { if (NT) __builtin_nontemporal_store(r[u], dst + e); else dst[e] = r[u]; }
        }
    }
}

template <int U>
__global__ __launch_bounds__(256) void read_k(double *out, const v2d *src, size_t n)
{
    const size_t chunk = (size_t)256 * U;
    v2d acc = {0, 0};
    for (size_t c = (size_t)blockIdx.x * chunk; c < n; c += (size_t)gridDim.x * chunk) {
        v2d r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t e = c + (size_t)u * 256 + threadIdx.x;
            r[u] = e < n ? __builtin_nontemporal_load(src + e) : v2d{0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += r[u];
    }
    if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

struct Alloc {
    void *ptr = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool vmm = false;
};

static Alloc alloc(size_t bytes, size_t chunk_mib)
{
    Alloc a;
    if (chunk_mib == 0) {
        a.bytes = bytes;
        CK(hipMalloc(&a.ptr, bytes));
        return a;
    }
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t chunk = chunk_mib << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    a.bytes = (bytes + chunk - 1) / chunk * chunk;
    a.vmm = true;
    CK(hipMemAddressReserve(&a.ptr, a.bytes, (size_t)1 << 30, nullptr, 0));
    for (size_t off = 0; off < a.bytes; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)a.ptr + off, chunk, 0, h, 0));
        a.handles.push_back(h);
    }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(a.ptr, a.bytes, &acc, 1));
    return a;
}

static void release(Alloc &a)
{
    if (!a.ptr) return;
    if (!a.vmm) { CK(hipFree(a.ptr)); }
    else {
        CK(hipMemUnmap(a.ptr, a.bytes));
        for (auto h : a.handles) CK(hipMemRelease(h));
        CK(hipMemAddressFree(a.ptr, a.bytes));
    }
    a = Alloc();
}

static float timed(hipStream_t s, int reps, const std::function<void()> &f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a, s));
        for (int q = 0; q < reps; ++q) f();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms / reps < best) best = ms / reps;
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return best;
}

int main(int argc, char **argv)
{
    const int ni = 4096, nk = 60, nj = 4096;
    const int ims = -31, ime = ims + 4160 - 1, jms = 0, jme = nj + 1, kms = 1, kme = nk + 1;
    const long idim = ime - ims + 1, kdim = kme - kms + 1, jdim = jme - jms + 1;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const double abytes = 8.0 * ni * nj * (11.0 * nk + 14);

    // ---- the box's streaming ceilings ----
    {
        const size_t nb = (size_t)4 << 30;
        void *src, *dst;
        CK(hipMalloc(&src, nb)); CK(hipMalloc(&dst, nb));
        CK(hipMemset(src, 1, nb)); CK(hipMemset(dst, 0, nb));
        const size_t n = nb / 16;
#define COPY(U, NT, G) { float ms = timed(s, 4, [&] { hipLaunchKernelGGL((copy_k<U, NT>), dim3(G), dim3(256), 0, s, (v2d *)dst, (const v2d *)src, n); }); \
        printf("{\"copy\": {\"unroll\": %d, \"nt\": %d, \"blocks\": %d, \"ms\": %.4f, \"GBps\": %.1f}}\n", U, (int)NT, G, ms, 2.0 * nb / ms / 1e6); fflush(stdout); }
        COPY(1, false, 2048) COPY(1, false, 8192) COPY(2, false, 2048) COPY(4, false, 2048) COPY(4, false, 4096) COPY(8, false, 2048)
        COPY(4, true, 2048) COPY(8, true, 2048) COPY(4, true, 1024) COPY(4, true, 4096) COPY(8, true, 1024) COPY(8, true, 4096) COPY(8, true, 512)
#define READ(U, G) { float ms = timed(s, 4, [&] { hipLaunchKernelGGL((read_k<U>), dim3(G), dim3(256), 0, s, (double *)dst, (const v2d *)src, n); }); \
        printf("{\"read\": {\"unroll\": %d, \"blocks\": %d, \"ms\": %.4f, \"GBps\": %.1f}}\n", U, G, ms, 1.0 * nb / ms / 1e6); fflush(stdout); }
        READ(4, 2048) READ(8, 2048) READ(8, 4096) READ(16, 2048)
        CK(hipFree(src)); CK(hipFree(dst));
    }

    // ---- the bench workload on differently obtained arrays ----
    std::vector<size_t> modes;
    for (int i = 1; i < argc; ++i) modes.push_back((size_t)atol(argv[i]));
    if (modes.empty()) modes = {0, 2, 64, 1024, 0, 2, 64, 1024};
    for (size_t mode : modes) {
        Alloc arr[AMT_F_COUNT];
        for (int f = 0; f < AMT_F_COUNT; ++f) {
            const int r = amt_field_rank(f);
            const size_t n = r == 3 ? (size_t)idim * kdim * jdim : r == 2 ? (size_t)idim * jdim : (size_t)kdim;
            arr[f] = alloc(n * 8, r == 3 ? mode : (mode ? 2 : 0));
            long fi = idim, fk = kdim, fj = jdim;
            if (amt_synth_fill_device(s, f, 8, arr[f].ptr, 1, fi, fk, fj, ims, kms - 1, jms, ni + 2, nk + 1, nj + 2) != 0) {
                fprintf(stderr, "fill: %s\n", amt_last_error());
                return 1;
            }
        }
        CK(hipStreamSynchronize(s));
#define P(f) (double *)arr[f].ptr
        auto call = [&] {
            int rc = amt_advance_mu_t_device_f64(s, 0, P(AMT_F_WW), P(AMT_F_WW_1), P(AMT_F_U), P(AMT_F_U_1), P(AMT_F_V), P(AMT_F_V_1),
                P(AMT_F_MU), P(AMT_F_MUT), P(AMT_F_MUAVE), P(AMT_F_MUTS), P(AMT_F_MUU), P(AMT_F_MUV), P(AMT_F_MUDF), P(AMT_F_T),
                P(AMT_F_T_1), P(AMT_F_T_AVE), P(AMT_F_FT), P(AMT_F_MU_TEND), AMT_SYNTH_RDX, AMT_SYNTH_RDY, AMT_SYNTH_DTS, AMT_SYNTH_EPSSM,
                P(AMT_F_DNW), P(AMT_F_FNM), P(AMT_F_FNP), P(AMT_F_RDNW), P(AMT_F_MSFUY), P(AMT_F_MSFVX_INV), P(AMT_F_MSFTX), P(AMT_F_MSFTY),
                0, 0, 0, 1, ni + 1, 1, nj + 1, nk + 1, ims, ime, jms, jme, kms, kme, 1, ni + 1, 1, nj + 1, 1, nk + 1);
            if (rc) { fprintf(stderr, "call: %s\n", amt_last_error()); exit(1); }
        };
        const float ms = timed(s, 4, call);
        printf("{\"mode_chunk_MiB\": %zu, \"ms\": %.3f, \"frac\": %.4f, \"base_u\": \"%p\"}\n", mode, ms, abytes / ms / 1e6 / 8000.0, arr[AMT_F_U].ptr);
        fflush(stdout);
        for (int f = 0; f < AMT_F_COUNT; ++f) release(arr[f]);
    }
    return 0;
}
