// amt_kernel_column.hip -- AMT_VARIANT_COLUMN: one lane per (i,j) column.
//
// First, simplest gfx950 kernel of the path (kept as the cross-check for the
// faster AMT_VARIANT_MARCH kernel).  Lanes of a wave64 are 64 consecutive i of one
// j-row, so every 3-D access of the wave is one contiguous 64*sizeof(T) segment
// (i-contiguous coalesced loads); each lane walks its own k-column twice:
//   pass 1  dvdxi(k) -> LDS column [k][lane], dmdt = sum_k dnw(k)*dvdxi(k)
//           (module_small_step_em.f90:140-149), then the 2-D mass update (:151-157)
//   pass 2  ww recurrence (:159-172), theta pre-update (:208-215), vertical flux
//           (:219-229) and flux-form theta update (:234-248), fused per level.
// The three Fortran phases fuse into one launch because a column never reads
// another column's outputs.  Each expression keeps the
// Fortran association and the file is compiled with -ffp-contract=off, so the
// results are bit-identical to the Fortran (no FMA, IEEE divide, the
// sequential k order of the dmdt sum and of the ww recurrence).
//
// No __syncthreads(): a lane only ever reads the LDS words it wrote itself.
//
// RECOMPUTE: the LDS column is nk * 64 * sizeof(T) bytes per wave -- 30 KB at 60 fp64 levels, 150 KB at 300, which leaves ONE wave per
// compute unit (AUTO falls back to this kernel beyond the march kernel's 240 / 264 levels: 9 % of the HBM roofline until r06).  With
// RECOMPUTE nothing is kept: pass 2 evaluates dvdxi(k) again from the same operands with the same expression -- the same bits -- at the
// price of reading u_1 and v_1 a second time (u and v are read by pass 2 anyway), and the occupancy is bounded by registers only.
// Measured (profiles/r06_tall_columns.md, 4096 columns wide): 300 fp64 levels 38.0 -> 8.4 ms (0.09 -> 0.41 of 8 TB/s), 60 levels 8.0 ->
// 6.3 ms, 20 levels 3.7 -> 4.3 ms: the launcher takes it where the LDS column is larger than 16 KB (more than 32 fp64 / 64 fp32 levels).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "amt_params.h"

// A workgroup is up to AMT_COLUMN_ROWS waves (blockDim.x / 64: the launcher's choice): wave w takes row (blockIdx / ntile_i) * rows + w of the tile.  The waves do not talk to each
// other; they share a compute unit and an XCD, so the rows j-1, j+1 a wave reads (t_1, v, v_1) are its siblings' own rows: fetched
// once into that L2 instead of once per row from HBM (profiles/r06_tall_columns.md).
#ifndef AMT_COLUMN_ROWS
#define AMT_COLUMN_ROWS 4
#endif

template <typename T, bool RECOMPUTE>
__global__ __launch_bounds__(64 * AMT_COLUMN_ROWS) void amt_column_kernel(const AmtParams<T> p, const int ntile_i)
{
    extern __shared__ __align__(16) unsigned char amt_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T *dv = reinterpret_cast<T *>(amt_smem) + (size_t)wave * (size_t)(p.nk > 0 ? p.nk : 1) * 64;     // [nk][64] per wave (unused with RECOMPUTE)

    const int tile = blockIdx.x % ntile_i;
    const int jrow = (blockIdx.x / ntile_i) * (int)(blockDim.x >> 6) + wave;
    const int ii   = tile * 64 + lane;                // zero-based memory i
    const int jj   = p.j0 + jrow;                     // zero-based memory j
    if (jj > p.j1 || ii < p.i0 || ii > p.i1) return;

    const long idim = p.idim;
    const long js   = p.jstride;
    const long c2   = (long)jj * idim + ii;           // (i,j) of 2-D arrays
    const long c3   = (long)jj * js + (long)p.k1 * idim + ii;   // (i,k=1,j)

    const T msftx = p.msftx[c2], msfty = p.msfty[c2];
    const T muu_i = p.muu[c2], muu_ip = p.muu[c2 + 1];
    const T msfuy_i = p.msfuy[c2], msfuy_ip = p.msfuy[c2 + 1];
    const T muv_j = p.muv[c2], muv_jp = p.muv[c2 + idim];
    const T mvx_j = p.msfvx_inv[c2], mvx_jp = p.msfvx_inv[c2 + idim];
    const T mu_tend = p.mu_tend[c2];
    const T rdx = p.rdx, rdy = p.rdy, dts = p.dts;
    const int nk = p.nk;
    const T *dnw = p.dnw + p.k1, *fnm = p.fnm + p.k1, *fnp = p.fnp + p.k1, *rdnw = p.rdnw + p.k1;
    // dnw[k-1] etc. below: index 0 is Fortran level 1

    // dvdxi(i,k,j), :141-146
    auto dvdxi = [&](long c) -> T {
        return msftx * msfty * (
              rdy * ( (p.v[c + js] + muv_jp * p.v_1[c + js] * mvx_jp)
                    - (p.v[c     ] + muv_j  * p.v_1[c     ] * mvx_j ) )
            + rdx * ( (p.u[c + 1] + muu_ip * p.u_1[c + 1] / msfuy_ip)
                    - (p.u[c    ] + muu_i  * p.u_1[c    ] / msfuy_i ) ));
    };

    // ---- pass 1: divergence and column integral (:140-149) ----
    T dmdt = T(0);
    for (int k = 0; k < nk; ++k) {
        const T d = dvdxi(c3 + (long)k * idim);
        if (!RECOMPUTE) dv[k * 64 + lane] = d;
        dmdt = dmdt + dnw[k] * d;
    }

    // ---- 2-D mass update (:151-157) ----
    {
        const T mu_old = p.mu[c2];
        const T mu_new = mu_old + dts * (dmdt + mu_tend);
        p.mu[c2]    = mu_new;
        p.mudf[c2]  = (dmdt + mu_tend);
        p.muts[c2]  = p.mut[c2] + mu_new;
        p.muave[c2] = T(.5) * ((T(1.) + p.epssm) * mu_new + (T(1.) - p.epssm) * mu_old);
    }
    if (nk < 1) return;

    // ---- pass 2: ww recurrence + theta, fused per level ----
    // ww_un : ww(k) of the recurrence (:161), before ww_1 is subtracted (:170)
    // wd_k  : wdtn(k) (:220,:227)
    T ww_un = p.ww[c3];
    T wout  = ww_un - p.ww_1[c3];
    T wd_k  = T(0);                                   // wdtn(i,1) = 0
    T t1_k  = p.t_1[c3];                              // t_1(i,k,j)
    for (int k = 0; k < nk; ++k) {                    // Fortran level k+1
        const long c = c3 + (long)k * idim;
        T wout_n = T(0), wd_n = T(0), t1_n = T(0);
        if (k + 1 < nk) {
            const T dv_k = RECOMPUTE ? dvdxi(c) : dv[k * 64 + lane];
            ww_un  = ww_un - dnw[k] * (dmdt + dv_k + mu_tend) / msfty;                 // :161
            wout_n = ww_un - p.ww_1[c + idim];                                         // :170
            t1_n   = p.t_1[c + idim];
            wd_n   = wout_n * (fnm[k + 1] * t1_n + fnp[k + 1] * t1_k);                 // :227
        }                                             // else wdtn(i,kde) = 0 (:221)
        p.ww[c] = wout;

        const T t_old = p.t[c];
        p.t_ave[c] = t_old;                                                            // :211
        const T t_b = t_old + msfty * dts * p.ft[c];                                   // :212
        p.t[c] = t_b - dts * msfty * (                                                 // :237-246
                    msftx * (
                        T(.5) * rdy *
                          ( p.v[c + js] * (p.t_1[c + js] + t1_k)
                          - p.v[c     ] * (t1_k + p.t_1[c - js]) )
                      + T(.5) * rdx *
                          ( p.u[c + 1] * (p.t_1[c + 1] + t1_k)
                          - p.u[c    ] * (t1_k + p.t_1[c - 1]) ) )
                  + rdnw[k] * (wd_n - wd_k) );
        wout = wout_n; wd_k = wd_n; t1_k = t1_n;
    }
}

template <typename T>
hipError_t amt_launch_column(hipStream_t stream, const AmtParams<T> &p)
{
    const int ni = p.i1 - p.i0 + 1, nj = p.j1 - p.j0 + 1;
    if (ni <= 0 || nj <= 0) return hipSuccess;
    const int tile_lo = p.i0 / 64, tile_hi = p.i1 / 64;
    const int ntile_i = tile_hi - tile_lo + 1;
    AmtParams<T> q = p;
    // shift the tile origin so that tile 0 is the first tile holding a window column
    // (lanes stay aligned to multiples of 64 elements from the row start)
    const long shift = (long)tile_lo * 64;
    q.ww += shift; q.mu += shift; q.muave += shift; q.muts += shift; q.mudf += shift;
    q.t += shift; q.t_ave += shift; q.ww_1 += shift; q.u += shift; q.u_1 += shift;
    q.v += shift; q.v_1 += shift; q.mut += shift; q.muu += shift; q.muv += shift;
    q.t_1 += shift; q.ft += shift; q.mu_tend += shift; q.msfuy += shift;
    q.msfvx_inv += shift; q.msftx += shift; q.msfty += shift;
    q.i0 -= (int)shift; q.i1 -= (int)shift;
    const size_t lds = (size_t)(p.nk > 0 ? p.nk : 1) * 64 * sizeof(T);      // of ONE wave's dvdxi column
    // one dvdxi column per lane in LDS: 160 KB hold 320 levels in fp64, 640 in fp32 (the header's level limit)
    if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
    auto grid_for = [&](int rows) { return (unsigned)((long)ntile_i * ((nj + rows - 1) / rows)); };
    // tall columns: nothing in LDS, dvdxi evaluated twice (see the head of this file); AMT_COLUMN_RECOMPUTE=0|1 forces either
    const char *force = getenv("AMT_COLUMN_RECOMPUTE");          // read per launch: a test flips it in-process
    const bool recompute = force && *force ? atoi(force) != 0 : lds > 16 * 1024;
    if (recompute) {
        hipLaunchKernelGGL((amt_column_kernel<T, true>), dim3(grid_for(AMT_COLUMN_ROWS)), dim3(64 * AMT_COLUMN_ROWS), 0, stream, q, ntile_i);
        return hipGetLastError();
    }
    // the LDS flavour: as many rows per workgroup as fit 64 KB of dvdxi columns (one where a single column is larger: the forced
    // AMT_COLUMN_RECOMPUTE=0 on tall columns)
    int rows = AMT_COLUMN_ROWS;
    while (rows > 1 && lds * rows > 64 * 1024) --rows;
    if (lds * rows > 64 * 1024) {
        // beyond 64 KB of dynamic LDS a kernel has to be allowed to (per device, once)
        static unsigned granted = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!(granted >> (dev & 31) & 1u)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(amt_column_kernel<T, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            granted |= 1u << (dev & 31);
        }
    }
    hipLaunchKernelGGL((amt_column_kernel<T, false>), dim3(grid_for(rows)), dim3(64 * rows), lds * rows, stream, q, ntile_i);
    return hipGetLastError();
}

template hipError_t amt_launch_column<float>(hipStream_t, const AmtParams<float> &);
template hipError_t amt_launch_column<double>(hipStream_t, const AmtParams<double> &);
