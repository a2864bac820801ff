#include <hip/hip_runtime.h>
extern "C" __global__ void k1(const float *a, const float *b, const float *c, float *o, int n, int flag)
{
    extern __shared__ float lds[];
    int i = threadIdx.x;
    float x0 = a[i], x1 = a[i + n], x2 = a[i + 2 * n];           // "P3 loads"
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float p0 = b[i], p1 = b[i + n], p2 = b[i + 2 * n];           // "prefetch"
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (flag > i) {
        float y = lds[i] + x0;                                    // should need vmcnt(5)
        y += x1; y += x2;
        o[i] = y;
    }
    o[i + n] = p0 + p1 + p2;
}
extern "C" __global__ void k2(const float *a, const float *b, const float *c, float *o, int n, int flag)
{
    extern __shared__ float lds[];
    int i = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(c + i), (__attribute__((address_space(3))) void *)lds, 4, 0, 0);
    float x0 = a[i], x1 = a[i + n], x2 = a[i + 2 * n];           // "P3 loads"
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float p0 = b[i], p1 = b[i + n], p2 = b[i + 2 * n];           // "prefetch"
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (flag > i) {
        float y = lds[i] + x0;
        y += x1; y += x2;
        o[i] = y;
    }
    o[i + n] = p0 + p1 + p2;
}
