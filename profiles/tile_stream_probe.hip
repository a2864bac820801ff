// profiles/tile_stream_probe.hip -- what does THIS BOX stream for the march kernel's ACCESS SHAPE, with nothing else in the way?
//
// A measurement aid, not part of the product (task 3: "show the amt_calib_stream_rate-style ceiling for
// 32-column tiles").  It reproduces the memory side of amt_march_kernel and nothing else -- no LDS staging, no barriers,
// no chains, no arithmetic beyond one add per loaded value:
//   * eight input arrays and three output arrays in WRF's (i, k, j) layout, i fastest;
//   * a workgroup owns one i tile of TC = (64 / HL) * VW columns, ALL levels, and marches through a block of j rows;
//   * a wave's 64 lanes are HL level groups of 64 / HL lanes; a lane owns VW adjacent columns and KPT consecutive levels,
//     so every wave-level access is HL contiguous runs of TC * W bytes, one level row apart -- 512 bytes for the 60-level
//     shapes (HL = 1), 256 bytes for the level-group shapes (HL = 2), 128 for HL = 4;
//   * per row a lane issues its 8 * KPT loads, then its 3 * KPT stores (the kernel's 8 reads + 3 writes per cell);
//   * the same XCD-aware workgroup order, 1024-thread workgroups, one per CU (dynamic LDS is requested to pin that).
// Prints the sustained rate per shape, and next to it a flat stream of the same bytes (one long run per workgroup).
//
//   hipcc -O3 --offload-arch=gfx950 -o profiles/tile_stream_probe profiles/tile_stream_probe.hip
//   profiles/tile_stream_probe [NI NJ]      (defaults 4096 2048)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T, int VW> struct VecOf { typedef T type __attribute__((ext_vector_type(VW))); };
template <typename T> struct VecOf<T, 1> { typedef T type; };

template <typename T, int VW, int KPT, int HL, bool FLAT>
__global__ __launch_bounds__(1024) void probe(const T *const *in, T *const *out, int idim, int kdim, int nk, int ntile, int jrows, int nj, int nwg)
{
    extern __shared__ unsigned char pin_lds[];
    constexpr int TI = 64 / HL, TC = TI * VW, LW = KPT * HL;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nx = 8, q = nwg / nx, r = nwg % nx;
    const int x = blockIdx.x % nx, y = blockIdx.x / nx;
    const int lid = x * q + (x < r ? x : r) + y;
    const int tile = lid % ntile, jblk = lid / ntile;
    const int ja = 1 + jblk * jrows, jb = min(ja + jrows - 1, nj);
    const int il = lane % TI, h = lane / TI;
    const int kf = w * LW + h * KPT;
    if (kf >= nk) return;
    if (pin_lds[0] == 123 && idim < 0) out[0][0] = T(1);     // keeps the LDS request alive
    typedef typename VecOf<T, VW>::type V;
    const size_t js = (size_t)idim * kdim;
    for (int j = ja; j <= jb; ++j) {
        V acc[KPT];
        size_t e[KPT];
#pragma unroll
        for (int m = 0; m < KPT; ++m) {
            const int k = min(kf + m, nk - 1);
            if (FLAT) {
                // the same bytes as one long run per (workgroup, row): levels of the tile laid end to end
                e[m] = ((size_t)j * ntile + tile) * ((size_t)nk * TC) + (size_t)k * TC + (size_t)il * VW;
            } else {
                e[m] = (size_t)j * js + (size_t)k * idim + 32 + (size_t)tile * TC + (size_t)il * VW;
            }
        }
        V v[8][KPT];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int m = 0; m < KPT; ++m) v[a][m] = *reinterpret_cast<const V *>(in[a] + e[m]);
#pragma unroll
        for (int m = 0; m < KPT; ++m) {
            acc[m] = v[0][m];
#pragma unroll
            for (int a = 1; a < 8; ++a) acc[m] += v[a][m];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int m = 0; m < KPT; ++m)
                if (kf + m < nk) *reinterpret_cast<V *>(out[a] + e[m]) = acc[m];
    }
}

template <typename T, int VW, int KPT, int HL>
static void run(const char *label, int ni, int nk, int nj, size_t lds)
{
    constexpr int TC = (64 / HL) * VW, LW = KPT * HL;
    const int idim = ni + 64, kdim = nk + 1, jdim = nj + 2;
    const size_t n = (size_t)idim * kdim * jdim;
    std::vector<T *> h_in(8), h_out(3);
    for (auto &p : h_in) { CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); }
    for (auto &p : h_out) { CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); }
    T **d_in, **d_out;
    CK(hipMalloc(&d_in, 8 * sizeof(T *))); CK(hipMalloc(&d_out, 3 * sizeof(T *)));
    CK(hipMemcpy(d_in, h_in.data(), 8 * sizeof(T *), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_out, h_out.data(), 3 * sizeof(T *), hipMemcpyHostToDevice));
    const int ntile = ni / TC, jrows = 64, njblk = (nj + jrows - 1) / jrows, nwg = ntile * njblk;
    const int nwav = (nk + LW - 1) / LW;
    const double bytes = 11.0 * sizeof(T) * ni * (double)nk * nj;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best[2] = {0, 0};
    for (int flat = 0; flat < 2; ++flat) {
        auto k = flat ? probe<T, VW, KPT, HL, true> : probe<T, VW, KPT, HL, false>;
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        std::vector<float> ms;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(nwg), dim3(nwav * 64), lds, 0, (const T *const *)d_in, (T *const *)d_out, idim, kdim, nk, ntile, jrows, nj, nwg);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        best[flat] = bytes / (ms[ms.size() / 2] * 1e-3) / 1e9;
    }
    printf("%-44s %4d B runs, %2d waves, %3zu KB LDS: tile-shaped %7.1f GB/s   flat %7.1f GB/s   ratio %.3f\n", label, (int)(TC * sizeof(T)), nwav,
           lds >> 10, best[0], best[1], best[0] / best[1]);
    fflush(stdout);
    for (auto p : h_in) CK(hipFree(p));
    for (auto p : h_out) CK(hipFree(p));
    CK(hipFree(d_in)); CK(hipFree(d_out));
}

int main(int argc, char **argv)
{
    const int ni = argc > 2 ? atoi(argv[1]) : 4096, nj = argc > 2 ? atoi(argv[2]) : 2048;
    printf("# %d columns x NK levels x %d rows, 8 loads + 3 stores per cell, 64-row blocks, one 1024-thread workgroup per CU\n", ni, nj);
    const bool sweep = argc > 3;
    if (!sweep) {
        run<double, 1, 4, 1>("fp64 60 levels (1,4,1): headline shape", ni, 60, nj, 120 << 10);
        run<double, 1, 3, 2>("fp64 80 levels (1,3,2)", ni, 80, nj, 100 << 10);
        run<double, 1, 4, 2>("fp64 80 levels (1,4,2)", ni, 80, nj, 100 << 10);
        run<double, 1, 3, 1>("fp64 45 levels (1,3,1)", ni, 45, nj, 100 << 10);
        run<double, 1, 4, 4>("fp64 160 levels (1,4,4)", ni, 160, nj / 2, 100 << 10);
        run<float, 2, 4, 1>("fp32 60 levels (2,4,1)", 2 * ni, 60, nj, 120 << 10);
        run<float, 2, 3, 2>("fp32 80 levels (2,3,2): configs[4] shape", 2 * ni, 80, nj, 100 << 10);
        run<float, 2, 4, 2>("fp32 80 levels (2,4,2)", 2 * ni, 80, nj, 100 << 10);
        run<float, 1, 4, 1>("fp32 60 levels (1,4,1): one column per lane", 2 * ni, 60, nj, 120 << 10);
    } else {
        // how the rate of the bare access shape depends on levels per lane and waves (any 4th argument)
        run<float, 2, 3, 2>("fp32 (2,3,2) 84 levels: 14 full waves", 2 * ni, 84, nj, 100 << 10);
        run<float, 2, 3, 2>("fp32 (2,3,2) 78 levels: 13 full waves", 2 * ni, 78, nj, 100 << 10);
        run<float, 2, 3, 2>("fp32 (2,3,2) 60 levels: 10 waves", 2 * ni, 60, nj, 100 << 10);
        run<float, 2, 3, 2>("fp32 (2,3,2) 90 levels: 15 waves", 2 * ni, 90, nj, 100 << 10);
        run<float, 2, 2, 2>("fp32 (2,2,2) 60 levels: 15 waves", 2 * ni, 60, nj, 100 << 10);
        run<float, 2, 4, 2>("fp32 (2,4,2) 80 levels: 10 waves", 2 * ni, 80, nj, 100 << 10);
        run<float, 2, 4, 2>("fp32 (2,4,2) 120 levels: 15 waves", 2 * ni, 120, nj / 2, 100 << 10);
        run<float, 2, 6, 2>("fp32 (2,6,2) 84 levels: 7 waves", 2 * ni, 84, nj, 100 << 10);
        run<float, 2, 8, 2>("fp32 (2,8,2) 80 levels: 5 waves", 2 * ni, 80, nj, 100 << 10);
        run<float, 2, 5, 2>("fp32 (2,5,2) 80 levels: 8 waves", 2 * ni, 80, nj, 100 << 10);
        run<float, 2, 4, 1>("fp32 (2,4,1) 60 levels: 15 waves, 512 B runs", 2 * ni, 60, nj, 120 << 10);
        run<float, 2, 3, 1>("fp32 (2,3,1) 45 levels: 15 waves, 512 B runs", 2 * ni, 45, nj, 120 << 10);
        run<float, 2, 6, 1>("fp32 (2,6,1) 60 levels: 10 waves, 512 B runs", 2 * ni, 60, nj, 120 << 10);
        run<float, 2, 8, 1>("fp32 (2,8,1) 80 levels: 10 waves, 512 B runs", 2 * ni, 80, nj, 120 << 10);
        run<double, 1, 3, 2>("fp64 (1,3,2) 84 levels: 14 full waves", ni, 84, nj, 100 << 10);
        run<double, 1, 4, 2>("fp64 (1,4,2) 80 levels: 10 waves", ni, 80, nj, 100 << 10);
        run<double, 1, 8, 1>("fp64 (1,8,1) 80 levels: 10 waves, 512 B runs", ni, 80, nj, 120 << 10);
        run<double, 1, 6, 1>("fp64 (1,6,1) 60 levels: 10 waves, 512 B runs", ni, 60, nj, 120 << 10);
    }
    return 0;
}
