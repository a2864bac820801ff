#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ src, double* __restrict__ dst, int lev_elems)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double* L = reinterpret_cast<double*>(smem);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* g = reinterpret_cast<const char*>(src) + (size_t)(w * 2 + (lane >> 5)) * lev_elems * 8 + (lane & 31) * 16;
    if (lane != 5)   // a masked lane must leave its 16 bytes untouched
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(L + w * 128), 16, 0, 0);
    __syncthreads();
    dst[threadIdx.x * 2] = L[threadIdx.x * 2];
    dst[threadIdx.x * 2 + 1] = L[threadIdx.x * 2 + 1];
}
int main()
{
    const int nw = 4, lev = 100;
    std::vector<double> h(nw * 2 * lev);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1000.0 + i;
    double *s, *d;
    hipMalloc(&s, h.size() * 8); hipMalloc(&d, nw * 128 * 8);
    hipMemcpy(s, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(d, 0, nw * 128 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(nw * 64), nw * 128 * 8, 0, s, d, lev);
    std::vector<double> o(nw * 128);
    hipMemcpy(o.data(), d, o.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0, masked_ok = 0;
    for (int w = 0; w < nw; ++w)
        for (int l = 0; l < 2; ++l)
            for (int i = 0; i < 64; ++i) {
                const double want = h[(w * 2 + l) * lev + i];
                const double got = o[w * 128 + l * 64 + i];
                const bool masked = (l == 0 && (i == 10 || i == 11));
                if (masked) { masked_ok += (got != want); continue; }
                if (got != want) { if (bad < 5) printf("w%d l%d i%d got %f want %f\n", w, l, i, got, want); ++bad; }
            }
    printf("bad=%d masked_untouched=%d (expect 0 and 8)\n", bad, masked_ok);
    return bad != 0;
}
