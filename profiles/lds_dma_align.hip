#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
// global_load_lds_dwordx4 from a source that is only 4- or 8-byte aligned: does it work, is it fast?
__global__ void k(const char *src, int shift, float *out, int n16)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    for (int it = 0; it < n16; ++it) {
        const char *g = src + shift + (size_t)blockIdx.x * 1024 * n16 + (size_t)it * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(smem + it * 1024), 16, 0, 0);
    }
    __syncthreads();
    const float *s = reinterpret_cast<const float *>(smem);
    for (int it = 0; it < n16; ++it)
        for (int q = 0; q < 4; ++q)
            out[((size_t)blockIdx.x * n16 + it) * 256 + lane * 4 + q] = s[it * 256 + lane * 4 + q];
}
__global__ void kt(const char *src, int shift, float *out, int n16, int reps)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int it = 0; it < n16; ++it) {
            const char *g = src + shift + ((size_t)(blockIdx.x * reps + r) * 4 + w) * 1024 * n16 + (size_t)it * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(smem + (w * n16 + it) * 1024), 16, 0, 0);
        }
        __syncthreads();
        acc += reinterpret_cast<const float *>(smem)[threadIdx.x];
        __syncthreads();
    }
    if (acc == 12345.f) out[0] = acc;
}
int main()
{
    const int nb = 64, n16 = 8;
    const size_t bytes = (size_t)nb * n16 * 1024 + 64;
    char *h = (char *)malloc(bytes), *d; float *o, *ho = (float *)malloc((size_t)nb * n16 * 1024);
    for (size_t i = 0; i < bytes / 4; ++i) ((float *)h)[i] = (float)i;
    hipMalloc(&d, bytes); hipMalloc(&o, (size_t)nb * n16 * 1024);
    hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    for (int shift : {0, 8, 4, 12}) {
        hipMemset(o, 0, (size_t)nb * n16 * 1024);
        hipLaunchKernelGGL(k, dim3(nb), dim3(64), n16 * 1024, 0, d, shift, o, n16);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(ho, o, (size_t)nb * n16 * 1024, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)nb * n16 * 256; ++i) if (ho[i] != (float)(i + shift / 4)) ++bad;
        printf("shift %2d: %s, %zu wrong of %zu\n", shift, hipGetErrorString(e), bad, (size_t)nb * n16 * 256);
    }
    // bandwidth: 2 GiB streamed through LDS-DMA, aligned vs misaligned
    const size_t big = (size_t)2 << 30;
    char *D; hipMalloc(&D, big + 4096); hipMemset(D, 1, big + 4096);
    const int reps = 16, n = 8; const int blocks = (int)(big / ((size_t)reps * 4 * 1024 * n));
    for (int shift : {0, 8, 4, 0, 8, 4}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(kt, dim3(blocks), dim3(256), 4 * n * 1024, 0, D, shift, o, n, reps);
        hipEventRecord(a);
        for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(kt, dim3(blocks), dim3(256), 4 * n * 1024, 0, D, shift, o, n, reps);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("stream 2 GiB by LDS-DMA, source shift %d: %.3f ms = %.2f TB/s\n", shift, ms / 3, big / (ms / 3) / 1e9);
    }
    return 0;
}
