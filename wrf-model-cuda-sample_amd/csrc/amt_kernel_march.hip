// amt_kernel_march.hip -- AMT_VARIANT_MARCH: (i,k)-cell lanes marching in j.
//
// Why: the column kernel reads v, v_1 twice (rows j and j+1) and t_1 three times
// (j-1, j, j+1) from beyond L2 -- rocprofv3 shows 1.97x the compulsory read bytes
// (profiles/r01_column_pmc.json).  This kernel reads every input element once per sweep:
//
//  * A workgroup owns one 64-wide i-tile (memory aligned), ALL levels, and a block
//    of consecutive j rows that it marches through.  Lane = i (so every 3-D access
//    of a wave is one contiguous 64*sizeof(T) run: i-contiguous coalesced loads).
//    NC "cell" waves own KPT consecutive levels each (wave w: levels w*KPT+1 .. (w+1)*KPT);
//    one more "column" wave owns everything that is per column (i,j) rather than per
//    cell: the two sequential k chains, the 2-D mass update, and the staging of the
//    2-D input rows (wave specialisation; 15 + 1 waves for NK = 60).
//  * The j-direction face fluxes  v(j)+muv(j)*v_1(j)*msfvx_inv(j)  and
//    v(j)*(t_1(j)+t_1(j-1))  are carried in registers from one row to the next (the face
//    j+1 of row j IS the face j of row j+1: the same expression on the same operands,
//    so carrying it is bit-exact); the t_1 row is carried in LDS (with its i halo), which
//    also serves t_1(i-1), t_1(i+1) and t_1(k-1).  v, v_1, t_1 are read once.
//  * The k-dependencies go through LDS (k-column staging), four barriers per j row:
//      cell waves   AP[k][lane] = dnw(k)*dvdxi(i,k)                        (:142-147)
//      -- barrier 1 --
//      column wave  dmdt = sum_k AP[k] in the Fortran's sequential k order -> DM[lane]
//      -- barrier 2 --
//      cell waves   AP[k][lane] <- dnw(k)*(dmdt+dvdxi(k)+mu_tend)/msfty  (:161, one divide
//                   per cell, own slots); column wave: the 2-D mass update (:151-157)
//      -- barrier 3 --
//      column wave  AP[k] <- ww(k); ww(k+1) = ww(k) - AP[k], sequential  (:161)
//      -- barrier 4 --
//      cell waves   ww - ww_1 (:170), wdtn (:220-227), theta update (:211-212, :237-246)
//    A column's chains are summed exactly once, in order: bit-exact and no redundant LDS
//    traffic (an earlier version let every wave redo both chains: LDS-bandwidth bound).
//  * Level counts that do not fill the cell waves: the last wave's missing levels are virtual
//    (clamped loads of its last real level, no stores) -- no per-level branches.
//  * Expressions keep the Fortran association; built with -ffp-contract=off.
//  * Two flavours of the same kernel: amt_march_kernel (general) and amt_march_dma_kernel, which
//    additionally prefetches the next row's t_1 and v through LDS-DMA (see its header below).
//
// Reference semantics: module_small_step_em.f90:112-172 (mu, ww), :208-215 and
// :217-250 (theta); the fusion of the three Fortran phases is legal because a
// column never reads another column's outputs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "amt_params.h"

static int amt_env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

#ifndef AMT_NT_STORE
#define AMT_NT_STORE 0
#endif
// Cache policy of the streams, measured in one process on the same arrays (profiles/ab_libs.py,
// 4096x60x4096 fp64, +-0.01 ms): nt loads of t, ft, ww_1 (each read exactly once) 16.28 -> 16.06 ms,
// every one of the three contributes; nt also on u, u_1 (each line is read by two overlapping
// loads) +0.4 ms; nt on the LDS-DMA loads +0.3 ms; nt stores +0.1 ms.
#ifndef AMT_NT_LOAD
#define AMT_NT_LOAD 1   /* 1: t, ft, ww_1 with the nt policy; 2: u, u_1 too */
#endif
#ifndef AMT_NT_DMA
#define AMT_NT_DMA 0    /* cache-policy bits of the bulk LDS-DMA loads (2 = nt) */
#endif
#ifndef AMT_COL_PRIO
#define AMT_COL_PRIO 0
#endif
#ifndef AMT_CHAIN
#define AMT_CHAIN 10    /* LDS reads kept in flight by the sequential k chains */
#endif

struct AmtMarchGrid {
    int ntile_i;     // number of 64-wide i tiles that hold window columns
    int tile_lo;     // first such tile
    int jrows;       // rows per workgroup
    int njblk;       // number of j blocks
    int nwg;         // ntile_i * njblk
    unsigned long long *stamps;   // diagnostic instantiation only: 8 cycle sums per wave
};

// Uniform-base addressing: every global access is  (wave-uniform pointer in SGPRs) +
// (32-bit per-lane byte offset) [+ immediate], i.e. `global_load ... v_off, s[base:base+1]`.
// 64-bit per-lane addresses would cost two VGPRs per distinct address and spill.
template <typename T>
__device__ __forceinline__ T amt_ld(const T *ubase, unsigned voff)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(ubase) + voff);
}
template <typename T>
__device__ __forceinline__ void amt_st(T *ubase, unsigned voff, T x)
{
    *reinterpret_cast<T *>(reinterpret_cast<char *>(ubase) + voff) = x;
}
// once-read inputs (level = which AMT_NT_LOAD setting turns the nt policy on for this stream)
template <int LEVEL, typename T>
__device__ __forceinline__ T amt_ld_stream(const T *ubase, unsigned voff)
{
#if AMT_NT_LOAD
    if (AMT_NT_LOAD >= LEVEL)
        return __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(ubase) + voff));
#endif
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(ubase) + voff);
}
// streaming store: the three 3-D outputs are written once and not read again in the sweep
template <typename T>
__device__ __forceinline__ void amt_st_stream(T *ubase, unsigned voff, T x)
{
#if AMT_NT_STORE
    __builtin_nontemporal_store(x, reinterpret_cast<T *>(reinterpret_cast<char *>(ubase) + voff));
#else
    *reinterpret_cast<T *>(reinterpret_cast<char *>(ubase) + voff) = x;
#endif
}

constexpr int AMT_TW = 66;    // LDS row buffers: 64 lanes + left/right halo
constexpr int AMT_N2D = 7;    // staged 2-D rows: msftx msfty muu msfuy muv' msfvx_inv' mu_tend

// FULL: nk is a multiple of KPT (every cell wave owns exactly KPT levels; no per-level guards).
// STAMP: diagnostic instantiation with s_memtime stamps per phase (never used for timings).
template <typename T, int KPT, bool FULL, bool STAMP>
__global__ __launch_bounds__(KPT >= 15 ? 320 : KPT >= 8 ? 704 : 1024) void amt_march_kernel(const AmtParams<T> p, const AmtMarchGrid g)
{
    extern __shared__ __align__(16) unsigned char amt_smem[];
    const int nk = p.nk;
    constexpr int TW = AMT_TW, N2D = AMT_N2D;
    // level rows of the LDS buffers: nk, or whole cell waves when nk % KPT != 0 (the last wave's
    // missing levels are virtual: clamped loads of its last real level, never stored -- see the
    // LDS-DMA flavour below)
    const int nkr = FULL ? nk : ((int)(blockDim.x >> 6) - 1) * KPT;
    // AP[k][lane] goes through three lives per row: dnw(k)*dvdxi(i,k), the term of dmdt (P1 ..
    // barrier 2); the ww increment of level k (barrier 2 .. 3); ww(k) of the recurrence :161
    // (barrier 4 .. P3).  A cell wave only ever reads its OWN slots after barrier 2, so the next
    // row's P1 may overwrite them without another barrier; the ww of the level above its last one
    // is rebuilt from its own last increment (the same subtraction the column wave performs).
    T *AP = reinterpret_cast<T *>(amt_smem);      // [nkr][64]
    T *T1 = AP + (size_t)nkr * 64;                // [2][nkr][66] t_1 of row j / row j+1 (+ i halo)
    T *D2 = T1 + (size_t)2 * nkr * TW;            // [2][N2D][66] 2-D inputs of row j / row j+1
    T *DM = D2 + (size_t)2 * N2D * TW;            // [64] dmdt of the row
    T *S1 = DM + 64;                              // dnw | fnm | fnp | rdnw, nkr entries each
    const T *s_dnw = S1, *s_fnm = S1 + nkr, *s_fnp = S1 + 2 * nkr, *s_rdnw = S1 + 3 * nkr;
    const int t1buf = nkr * TW, d2buf = N2D * TW;

    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
    const int nwav = (int)(blockDim.x >> 6);
    const int nc   = nwav - 1;                                           // cell waves 0..nc-1
    const bool colw = (w == nc);                                         // the column wave
    const unsigned vo = (unsigned)lane * (unsigned)sizeof(T);            // the only per-lane offset

    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int slot) {
        if (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (slot >= 0) st_acc[slot] += now - st_prev;
            st_prev = now;
        }
    };

    // XCD-aware logical workgroup id: blocks b, b+8, b+16 ... share an XCD (round-robin
    // dispatch), so give each XCD a contiguous run of logical ids: neighbouring i-tiles
    // of one j block then run on one XCD at about the same time and share the tile-edge
    // cache lines in that XCD's L2.  Speed only, never correctness.
    int lid;
    {
        const int nx = 8, q = g.nwg / nx, r = g.nwg % nx;
        const int x = blockIdx.x % nx, y = blockIdx.x / nx;
        lid = x * q + (x < r ? x : r) + y;        // XCD x owns q (+1 if x < r) consecutive ids
    }
    const int tile = g.tile_lo + lid % g.ntile_i;
    const int jblk = lid / g.ntile_i;

    for (int e = threadIdx.x; e < 4 * nkr; e += blockDim.x) {
        const int which = e / nkr, k = e % nkr;
        const T *src = which == 0 ? p.dnw : which == 1 ? p.fnm : which == 2 ? p.fnp : p.rdnw;
        S1[e] = src[p.k1 + (k < nk ? k : nk - 1)];
    }

    const int ii   = tile * 64 + lane;
    const bool act = (ii >= p.i0) && (ii <= p.i1);               // column is in the window
    const bool t1ok = (ii >= p.i0 - 1) && (ii <= p.i1 + 1);      // its t_1 is read by a window column
    const bool edge = act && (lane == 0 || lane == 63);          // loads the tile's i halo of t_1
    const bool inmem = ii < p.idim;                              // lane is inside the memory row
    const bool halo_r = (lane == 0) && (ii + 64 < p.idim);       // lane 0 also fetches element i+64
    const unsigned eoff = (lane == 0) ? 0u : 2u * (unsigned)sizeof(T);   // from base-1: i-1 / i+1
    const int ehalo = (lane == 0) ? 0 : TW - 1;                  // halo slot in a row buffer
    const int ja   = p.j0 + jblk * g.jrows;
    const int jb   = (ja + g.jrows - 1 < p.j1) ? ja + g.jrows - 1 : p.j1;

    const long idim = p.idim, js = p.jstride;
    const unsigned lev = (unsigned)idim * (unsigned)sizeof(T);      // byte step of one level
    const unsigned row3 = (unsigned)js * (unsigned)sizeof(T);       // byte step of one j row (3-D)
    const unsigned row2 = lev;                                      // byte step of one j row (2-D)
    const long e2 = (long)ja * idim + (long)tile * 64;              // (lane 0 of the tile, row ja)

    // The 2-D inputs of a row are fetched ONCE per workgroup, one row ahead, by the column
    // wave and handed to the cell waves through LDS.  Slot order of D2:
    //   0 msftx(j) 1 msfty(j) 2 muu(j) 3 msfuy(j) 4 muv(j+1) 5 msfvx_inv(j+1) 6 mu_tend(j)
    auto d2_src = [&](int q) -> const T * {
        switch (q) {
        case 0: return p.msftx + e2;
        case 1: return p.msfty + e2;
        case 2: return p.muu + e2;
        case 3: return p.msfuy + e2;
        case 4: return p.muv + e2 + idim;
        case 5: return p.msfvx_inv + e2 + idim;
        default: return p.mu_tend + e2;
        }
    };

    if (colw) {
        // =====================================================================
        // column wave
        // =====================================================================
        const T *wwin_b = p.ww + (long)ja * js + (long)p.k1 * idim + (long)tile * 64;   // level 1
        const T *mut_b = p.mut + e2;
        T *mu_b = p.mu + e2, *mudf_b = p.mudf + e2, *muts_b = p.muts + e2, *muave_b = p.muave + e2;
        const T dts = p.dts;
        if (AMT_COL_PRIO) __builtin_amdgcn_s_setprio(3);   // the other 15 waves wait on this one's chains

        // prologue: 2-D row ja into D2 buffer 0
#pragma unroll
        for (int q = 0; q < N2D; ++q) {
            const T *src = d2_src(q);
            if (inmem) D2[q * TW + 1 + lane] = amt_ld(src, vo);
            if (halo_r) D2[q * TW + TW - 1] = amt_ld(src + 64, vo);
        }
        __syncthreads();                           // S1, T1[0], D2[0] staged

        unsigned o3 = vo, o2 = vo;
        for (int jj = ja; jj <= jb; ++jj, o3 += row3, o2 += row2) {
            const int par = (jj - ja) & 1;
            const T *D2c = D2 + par * d2buf;
            T *D2n = D2 + (par ^ 1) * d2buf;
            const bool more = (jj < jb);
            stamp(-1);
            // while the cell waves do P1: fetch the next 2-D row and this row's column inputs
            T d2v[N2D], d2h[N2D];
#pragma unroll
            for (int q = 0; q < N2D; ++q) { d2v[q] = T(0); d2h[q] = T(0); }
            if (more) {
#pragma unroll
                for (int q = 0; q < N2D; ++q) {
                    const T *src = d2_src(q);
                    if (inmem) d2v[q] = amt_ld(src, o2 + row2);
                    if (halo_r) d2h[q] = amt_ld(src + 64, o2 + row2);
                }
            }
            T ww1in = T(0), mu_old = T(0), mut_v = T(0);
            if (act) {
                ww1in = amt_ld(wwin_b, o3);                      // incoming ww(i,1,j)
                mu_old = amt_ld(mu_b, o2);
                mut_v = amt_ld(mut_b, o2);
            }
            const T mu_tend = D2c[6 * TW + 1 + lane];
            stamp(0);
            __syncthreads();                                     // 1: AP complete
            stamp(1);
            T dmdt = T(0);
            {                                                    // :147, sequential in k
                int k = 0;
                for (; k + AMT_CHAIN <= nk; k += AMT_CHAIN) {
                    T a[AMT_CHAIN];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) a[q] = AP[(k + q) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) dmdt = dmdt + a[q];
                }
                for (; k < nk; ++k) dmdt = dmdt + AP[k * 64 + lane];
            }
            DM[lane] = dmdt;
            stamp(2);
            __syncthreads();                                     // 2: DM published
            stamp(3);
            if (act) {                                           // :151-157
                const T mu_new = mu_old + dts * (dmdt + mu_tend);
                amt_st(mu_b, o2, mu_new);
                amt_st(mudf_b, o2, (dmdt + mu_tend));
                amt_st(muts_b, o2, mut_v + mu_new);
                amt_st(muave_b, o2, T(.5) * ((T(1.) + p.epssm) * mu_new + (T(1.) - p.epssm) * mu_old));
            }
            if (more) {                                          // hand the 2-D row j+1 over
#pragma unroll
                for (int q = 0; q < N2D; ++q) {
                    if (inmem) D2n[q * TW + 1 + lane] = d2v[q];
                    if (halo_r) D2n[q * TW + TW - 1] = d2h[q];
                }
            }
            stamp(4);
            __syncthreads();                                     // 3: AP holds the increments
            stamp(5);
            {                                                    // :161, sequential in k; AP[k] <- ww(k), the value BEFORE increment k
                T wwu = ww1in;
                int k = 0;
                for (; k + AMT_CHAIN <= nk; k += AMT_CHAIN) {
                    T b[AMT_CHAIN];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) b[q] = AP[(k + q) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) { AP[(k + q) * 64 + lane] = wwu; wwu = wwu - b[q]; }
                }
                for (; k < nk; ++k) { const T bk = AP[k * 64 + lane]; AP[k * 64 + lane] = wwu; wwu = wwu - bk; }
            }
            stamp(6);
            __syncthreads();                                     // 4: ww of the recurrence published
            stamp(7);
        }
    } else {
        // =====================================================================
        // cell waves
        // =====================================================================
        const int kf   = w * KPT;                      // my levels: zero-based kf .. kf+nlev-1
        const int nlev = FULL ? KPT : ((nk - kf < KPT) ? (nk - kf) : KPT);
        auto lv = [&](int m) { return FULL ? m : (m < nlev ? m : nlev - 1); };   // level a (virtual) slot loads
        const bool has_above = (kf + KPT < nk);        // zero-based level kf+KPT exists
        const T rdx = p.rdx, rdy = p.rdy, dts = p.dts;
        const T hrdy = T(.5) * rdy, hrdx = T(.5) * rdx;

        // Wave-uniform base pointers (SGPR pairs), fixed for the whole march: element
        // (lane 0 of the tile, my first level, row ja) of every 3-D array.  The row advance and
        // the level step go into a 32-bit per-lane byte offset (o3), which the launcher keeps
        // below 2^31.
        const long e3 = (long)ja * js + (long)(p.k1 + kf) * idim + (long)tile * 64;
        const T *u_b = p.u + e3, *u1_b = p.u_1 + e3, *ft_b = p.ft + e3, *ww1_b = p.ww_1 + e3;
        const T *vn_b = p.v + e3 + js, *v1n_b = p.v_1 + e3 + js, *t1n_b = p.t_1 + e3 + js;   // row j+1
        T *t_b = p.t + e3, *tave_b = p.t_ave + e3, *ww_b = p.ww + e3;

        // carried in registers from row to row (per owned level): the two j-face fluxes
        T vfm[KPT], vft[KPT];
#pragma unroll
        for (int m = 0; m < KPT; ++m) { vfm[m] = vft[m] = T(0); }

        // ---- prologue: j-face fluxes of row ja, t_1 row ja into LDS buffer 0 ----
        {
            T muv_j = T(0), mvx_j = T(0);
            if (act) { muv_j = amt_ld(p.muv + e2, vo); mvx_j = amt_ld(p.msfvx_inv + e2, vo); }
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                const unsigned om = vo + (unsigned)lv(m) * lev;
                const int K = kf + m;
                if (t1ok) {
                    const T tc = amt_ld(p.t_1 + e3, om);
                    T1[K * TW + 1 + lane] = tc;
                    if (act) {
                        const T vv = amt_ld(p.v + e3, om);
                        vfm[m] = vv + muv_j * amt_ld(p.v_1 + e3, om) * mvx_j;
                        vft[m] = vv * (tc + amt_ld(p.t_1 + e3 - js, om));
                    }
                }
                if (edge) T1[K * TW + ehalo] = amt_ld(p.t_1 + e3 - 1, om + eoff);
            }
        }
        __syncthreads();                               // S1, T1[0], D2[0] staged

        unsigned o3 = vo;                              // per-lane byte offset of the current row
        for (int jj = ja; jj <= jb; ++jj, o3 += row3) {
            T hf[KPT], tw[KPT], dv[KPT];
            T msfty = T(1), mu_tend = T(0), tw_above = T(0);
            const int par = (jj - ja) & 1;
            const T *T1c = T1 + par * t1buf;                         // t_1 row j   (read)
            T *T1n = T1 + (par ^ 1) * t1buf;                         // t_1 row j+1 (written, read next row)
            const T *D2c = D2 + par * d2buf;                         // 2-D row j   (read)
            stamp(-1);

            // ---------------- P1: per-cell work from pure inputs ----------------
            if (t1ok && !act) {                                      // the (at most two) columns beside the window
#pragma unroll
                for (int m = 0; m < KPT; ++m)
                    T1n[(kf + m) * TW + 1 + lane] = amt_ld(t1n_b, o3 + (unsigned)lv(m) * lev);
            }
            if (edge) {
#pragma unroll
                for (int m = 0; m < KPT; ++m)
                    T1n[(kf + m) * TW + ehalo] = amt_ld(t1n_b - 1, o3 + (unsigned)lv(m) * lev + eoff);
            }
            if (act) {
                const T msftx = D2c[0 * TW + 1 + lane];
                msfty = D2c[1 * TW + 1 + lane];
                const T mm = msftx * msfty;
                const T muu_i = D2c[2 * TW + 1 + lane], muu_ip = D2c[2 * TW + 2 + lane];
                const T msfuy_i = D2c[3 * TW + 1 + lane], msfuy_ip = D2c[3 * TW + 2 + lane];
                const T muv_p = D2c[4 * TW + 1 + lane], mvx_p = D2c[5 * TW + 1 + lane];
                mu_tend = D2c[6 * TW + 1 + lane];
                if (has_above) {
                    // wdtn at the level above my last one needs that level's t_1 pair (:227)
                    const int Ka = kf + KPT;
                    tw_above = s_fnm[Ka] * T1c[Ka * TW + 1 + lane] + s_fnp[Ka] * T1c[(Ka - 1) * TW + 1 + lane];
                }
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + (unsigned)lv(m) * lev;
                    const int K = kf + m;
                    const T vn = amt_ld(vn_b, om), v1n = amt_ld(v1n_b, om);
                    const T t1n = amt_ld(t1n_b, om);                       // t_1(i,k,j+1)
                    T1n[K * TW + 1 + lane] = t1n;
                    const T uu = amt_ld(u_b, om), uup = amt_ld(u_b + 1, om);
                    const T u1 = amt_ld(u1_b, om), u1p = amt_ld(u1_b + 1, om);
                    const T t1c = T1c[K * TW + 1 + lane], t1l = T1c[K * TW + lane], t1r = T1c[K * TW + 2 + lane];
                    // :142-146
                    const T vfm_n = vn + muv_p * v1n * mvx_p;
                    const T d = mm * ( rdy * (vfm_n - vfm[m])
                                     + rdx * ( (uup + muu_ip * u1p / msfuy_ip)
                                             - (uu  + muu_i  * u1  / msfuy_i ) ));
                    dv[m] = d;
                    AP[K * 64 + lane] = s_dnw[K] * d;            // the term of :147
                    // horizontal part of :237-245
                    const T vft_n = vn * (t1n + t1c);
                    hf[m] = msftx * ( hrdy * (vft_n - vft[m])
                                    + hrdx * ( uup * (t1r + t1c) - uu * (t1c + t1l) ) );
                    // fnm(k)*t_1(k) + fnp(k)*t_1(k-1) of :227 (unused for Fortran level 1)
                    const T t1km1 = (K > 0) ? T1c[(K > 0 ? K - 1 : 0) * TW + 1 + lane] : T(0);
                    tw[m] = s_fnm[K] * t1c + s_fnp[K] * t1km1;
                    vfm[m] = vfm_n; vft[m] = vft_n;              // the faces of row j+1
                }
            }
            stamp(0);
            __syncthreads();                                         // 1: AP complete
            stamp(1);

            // while the column wave sums dmdt: issue the loads that only P3 consumes
            T told[KPT], ftk[KPT], w1[KPT];
            T w1_above = T(0);
            if (act) {
                if (has_above) w1_above = amt_ld(ww1_b, o3 + (unsigned)KPT * lev);
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + (unsigned)lv(m) * lev;
                    told[m] = amt_ld_stream<1>(t_b, om);
                    ftk[m] = amt_ld_stream<1>(ft_b, om);
                    w1[m] = amt_ld_stream<1>(ww1_b, om);
                }
            }
            stamp(2);
            __syncthreads();                                         // 2: DM published
            stamp(3);
            T inc_last = T(0);                                      // my top level's increment (:161)
            if (act) {
                const T dmdt = DM[lane];
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const int K = kf + m;
                    const T inc = s_dnw[K] * (dmdt + dv[m] + mu_tend) / msfty;   // :161
                    AP[K * 64 + lane] = inc;
                    if (m == KPT - 1) inc_last = inc;
                }
            }
            stamp(4);
            __syncthreads();                                         // 3: AP holds the increments
            stamp(5);
            // nothing to do while the column wave runs the ww recurrence
            stamp(6);
            __syncthreads();                                         // 4: AP[k] = ww(k) of the recurrence
            stamp(7);

            // ---------------- P3: vertical flux, theta ----------------
            if (act) {
                T wwu = AP[kf * 64 + lane];                          // ww of :161 at my first level
                T wd_k = (kf == 0) ? T(0) : (wwu - w1[0]) * tw[0];   // wdtn(i,1) = 0 (:220)
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + (unsigned)lv(m) * lev;
                    const int K = kf + m;
                    const bool real = FULL || m < nlev;          // wave-uniform
                    const T wout = wwu - w1[m];                  // :170
                    if (real) amt_st_stream(ww_b, om, wout);
                    // wdtn at level K+1 (:221,:227)
                    T wd_n = T(0);
                    const T wwu_n = (m + 1 < KPT) ? AP[(m + 1 < KPT ? K + 1 : K) * 64 + lane] : wwu - inc_last;
                    if (m + 1 < KPT) {
                        wd_n = (wwu_n - w1[m + 1 < KPT ? m + 1 : 0]) * tw[m + 1 < KPT ? m + 1 : 0];
                        if (!FULL && K + 1 >= nk) wd_n = T(0);                            // wdtn(kde) = 0, :221
                    } else if (has_above) {
                        wd_n = (wwu_n - w1_above) * tw_above;
                    }
                    if (real) amt_st_stream(tave_b, om, told[m]);                                // :211
                    const T tb = told[m] + msfty * dts * ftk[m];                          // :212
                    if (real) amt_st_stream(t_b, om, tb - dts * msfty * ( hf[m] + s_rdnw[K] * (wd_n - wd_k) ));   // :237-246
                    wwu = wwu_n; wd_k = wd_n;
                }
            }
            // No barrier here.  What the next row's P1 writes (its own AP slots, and the T1 buffer
            // that was READ in this row's P1) is read by no other wave before barrier 1 of the next row.
        }
    }
    if (STAMP && g.stamps && lane == 0) {
        for (int q = 0; q < 8; ++q) g.stamps[((size_t)blockIdx.x * 16 + w) * 8 + q] = st_acc[q];
    }
}

// ---------------------------------------------------------------------------
// AMT_VARIANT_MARCH, LDS-DMA flavour (the fast path when the arrays allow it)
// ---------------------------------------------------------------------------
// Same mapping, same four barriers, same arithmetic as amt_march_kernel.  Difference: the rows
// j+2 of t_1 and v that the NEXT row's P1 needs are fetched by LDS-DMA (`global_load_lds_dwordx4`,
// no VGPR destination) right after barrier 1, i.e. while the column wave runs its chains and
// the cell waves would otherwise only wait -- the stamps of the plain kernel show the CU's memory
// queue saturated during P1 and nearly idle in the other 40 % of a row.  fp64 has no registers
// left for a classic prefetch (every such variant spilled); the DMA needs none.  What it needs
// is LDS: t_1 rows 64 wide (halo kept apart in TH) and one more [nk][64] buffer for v, paid for
// by single-buffering the 2-D rows.  Barriers 2 and 3 are LDS-only (inline asm): a
// __syncthreads() would drain the DMA (hipcc waits vmcnt(0) at a workgroup fence while an
// LDS-DMA is in flight); barrier 4 is a full one and is where the DMA must have landed.
// Requirements checked by the launcher (amt_march_dma_ok): rows are a multiple of 16 bytes, the
// DMA'd arrays are 16-byte aligned, nk % KPT == 0 and every cell wave owns whole DMA instructions
// (one instruction moves 64*16 bytes = 2 levels in fp64, 4 in fp32), and the LDS budget holds.
__device__ __forceinline__ void amt_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// XD: how many MORE input rows ride the DMA when LDS allows (fp32, or fp64 with nk <= ~44):
//   0: t_1, v   1: + v_1 (row j+2)   2: + u (row j+1, with its i+1 halo)   3: + u_1  -> no global load left in P1
// FULL: nk is a multiple of KPT.  Otherwise the last cell wave owns fewer real levels; its missing
// ones are VIRTUAL: they load the wave's last real level again (addresses clamped, wave-uniform),
// compute on the duplicate and are never stored, so that the code stays free of per-level branches
// (which would break the load batching) -- the LDS level buffers then hold nkr = waves*KPT rows.
template <typename T, int KPT, int XD, bool FULL>
__global__ __launch_bounds__(KPT >= 15 ? 320 : KPT >= 8 ? 704 : 1024) void amt_march_dma_kernel(const AmtParams<T> p, const AmtMarchGrid g)
{
    extern __shared__ __align__(16) unsigned char amt_smem[];
    const int nk = p.nk;
    constexpr int TW = AMT_TW, N2D = AMT_N2D;
    constexpr int EPL = 16 / (int)sizeof(T);      // elements per lane of one DMA instruction
    constexpr int LPL = 64 / EPL;                 // lanes per level row (64 elements)
    constexpr int LPI = 64 / LPL;                 // levels per DMA instruction
    static_assert(KPT % LPI == 0, "a cell wave must own whole DMA instructions");
    // AB[k][lane] goes through three lives per row: dvdxi(i,k) (P1 .. barrier 2), the ww increment
    // of level k (barrier 2 .. 3), ww(k) of the recurrence :161 (barrier 4 .. P3).  A cell wave
    // only ever reads its OWN slots after barrier 2, so the next row's P1 may overwrite them
    // without another barrier; the value it needs from the wave above is rebuilt from its own
    // last increment, kept in a register (the same subtraction the column wave performs).
    const int nkr = FULL ? nk : ((int)(blockDim.x >> 6) - 1) * KPT;   // level rows of the LDS buffers
    T *AB = reinterpret_cast<T *>(amt_smem);      // [nkr][64]
    T *T1 = AB + (size_t)nkr * 64;                // [2][nkr][64] t_1 rows (buffer = row parity)
    T *V  = T1 + (size_t)2 * nkr * 64;            // [nkr][64]   v of row j+1
    T *TH = V + (size_t)nkr * 64;                 // [2][nkr][2] i halo of the t_1 rows: left, right
    T *V1 = TH + (size_t)4 * nkr;                 // [nkr][64]   v_1 of row j+1            (XD >= 1)
    T *U  = V1 + (XD >= 1 ? (size_t)nkr * 64 : 0); // [nkr][64]  u of row j                (XD >= 2)
    T *U1 = U + (XD >= 2 ? (size_t)nkr * 64 : 0); // [nkr][64]   u_1 of row j              (XD >= 3)
    T *UH = U1 + (XD >= 3 ? (size_t)nkr * 64 : 0); // [2][nkr]   element i+64 of the u / u_1 rows (XD >= 2)
    T *D2 = UH + (XD >= 2 ? (size_t)2 * nkr : 0); // [N2D][66]  2-D inputs of the current row
    T *DM = D2 + (size_t)N2D * TW;                // [64]
    T *S1 = DM + 64;                              // dnw | fnm | fnp | rdnw, nkr entries each
    const T *s_dnw = S1, *s_fnm = S1 + nkr, *s_fnp = S1 + 2 * nkr, *s_rdnw = S1 + 3 * nkr;
    const int t1buf = nkr * 64, thbuf = nkr * 2;

    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwav = (int)(blockDim.x >> 6);
    const int nc   = nwav - 1;
    const bool colw = (w == nc);
    const unsigned vo = (unsigned)lane * (unsigned)sizeof(T);

    int lid;
    {
        const int nx = 8, q = g.nwg / nx, r = g.nwg % nx;
        const int x = blockIdx.x % nx, y = blockIdx.x / nx;
        lid = x * q + (x < r ? x : r) + y;
    }
    const int tile = g.tile_lo + lid % g.ntile_i;
    const int jblk = lid / g.ntile_i;

    for (int e = threadIdx.x; e < 4 * nkr; e += blockDim.x) {
        const int which = e / nkr, k = e % nkr;
        const T *src = which == 0 ? p.dnw : which == 1 ? p.fnm : which == 2 ? p.fnp : p.rdnw;
        S1[e] = src[p.k1 + (k < nk ? k : nk - 1)];
    }

    const int ii   = tile * 64 + lane;
    const bool act = (ii >= p.i0) && (ii <= p.i1);
    const bool inmem = ii < p.idim;
    const bool halo_r = (lane == 0) && (ii + 64 < p.idim);
    const int ja   = p.j0 + jblk * g.jrows;
    const int jb   = (ja + g.jrows - 1 < p.j1) ? ja + g.jrows - 1 : p.j1;

    const long idim = p.idim, js = p.jstride;
    const unsigned lev = (unsigned)idim * (unsigned)sizeof(T);
    const unsigned row3 = (unsigned)js * (unsigned)sizeof(T);
    const unsigned row2 = lev;
    const long e2 = (long)ja * idim + (long)tile * 64;

    auto d2_src = [&](int q) -> const T * {
        switch (q) {
        case 0: return p.msftx + e2;
        case 1: return p.msfty + e2;
        case 2: return p.muu + e2;
        case 3: return p.msfuy + e2;
        case 4: return p.muv + e2 + idim;
        case 5: return p.msfvx_inv + e2 + idim;
        default: return p.mu_tend + e2;
        }
    };

    if (colw) {
        // ===================== column wave (as in amt_march_kernel; D2 single-buffered) =====================
        const T *wwin_b = p.ww + (long)ja * js + (long)p.k1 * idim + (long)tile * 64;
        const T *mut_b = p.mut + e2;
        T *mu_b = p.mu + e2, *mudf_b = p.mudf + e2, *muts_b = p.muts + e2, *muave_b = p.muave + e2;
        const T dts = p.dts;
#pragma unroll
        for (int q = 0; q < N2D; ++q) {
            const T *src = d2_src(q);
            if (inmem) D2[q * TW + 1 + lane] = amt_ld(src, vo);
            if (halo_r) D2[q * TW + TW - 1] = amt_ld(src + 64, vo);
        }
        __syncthreads();                           // S1, T1, TH, V, D2 staged

        unsigned o3 = vo, o2 = vo;
        for (int jj = ja; jj <= jb; ++jj, o3 += row3, o2 += row2) {
            const bool more = (jj < jb);
            T d2v[N2D], d2h[N2D];
#pragma unroll
            for (int q = 0; q < N2D; ++q) { d2v[q] = T(0); d2h[q] = T(0); }
            if (more) {
#pragma unroll
                for (int q = 0; q < N2D; ++q) {
                    const T *src = d2_src(q);
                    if (inmem) d2v[q] = amt_ld(src, o2 + row2);
                    if (halo_r) d2h[q] = amt_ld(src + 64, o2 + row2);
                }
            }
            T ww1in = T(0), mu_old = T(0), mut_v = T(0);
            if (act) {
                ww1in = amt_ld(wwin_b, o3);
                mu_old = amt_ld(mu_b, o2);
                mut_v = amt_ld(mut_b, o2);
            }
            const T mu_tend = D2[6 * TW + 1 + lane];
            __syncthreads();                                     // 1: AP complete, D2/T1 row j no longer read
            T dmdt = T(0);
            {
                int k = 0;
                for (; k + AMT_CHAIN <= nk; k += AMT_CHAIN) {
                    T a[AMT_CHAIN];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) a[q] = s_dnw[k + q] * AB[(k + q) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) dmdt = dmdt + a[q];
                }
                for (; k < nk; ++k) dmdt = dmdt + s_dnw[k] * AB[k * 64 + lane];
            }
            DM[lane] = dmdt;
            if (more) {                                          // install the 2-D row j+1
#pragma unroll
                for (int q = 0; q < N2D; ++q) {
                    if (inmem) D2[q * TW + 1 + lane] = d2v[q];
                    if (halo_r) D2[q * TW + TW - 1] = d2h[q];
                }
            }
            amt_lds_barrier();                                   // 2: DM published
            if (act) {
                const T mu_new = mu_old + dts * (dmdt + mu_tend);
                amt_st(mu_b, o2, mu_new);
                amt_st(mudf_b, o2, (dmdt + mu_tend));
                amt_st(muts_b, o2, mut_v + mu_new);
                amt_st(muave_b, o2, T(.5) * ((T(1.) + p.epssm) * mu_new + (T(1.) - p.epssm) * mu_old));
            }
            amt_lds_barrier();                                   // 3: AB holds the increments
            {
                T wwu = ww1in;                                   // AB[k] <- ww(k), the value BEFORE increment k
                int k = 0;
                for (; k + AMT_CHAIN <= nk; k += AMT_CHAIN) {
                    T b[AMT_CHAIN];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) b[q] = AB[(k + q) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) { AB[(k + q) * 64 + lane] = wwu; wwu = wwu - b[q]; }
                }
                for (; k < nk; ++k) { const T bk = AB[k * 64 + lane]; AB[k * 64 + lane] = wwu; wwu = wwu - bk; }
            }
            __syncthreads();                                     // 4
        }
    } else {
        // ===================== cell waves =====================
        const int kf = w * KPT;
        const bool has_above = (kf + KPT < nk);
        const int nlev = FULL ? KPT : (nk - kf < KPT ? nk - kf : KPT);      // real levels of this wave (>= 1)
        auto lv = [&](int m) { return FULL ? m : (m < nlev ? m : nlev - 1); };   // level a (virtual) slot loads
        const T rdx = p.rdx, rdy = p.rdy, dts = p.dts;
        const T hrdy = T(.5) * rdy, hrdx = T(.5) * rdx;
        const long e3 = (long)ja * js + (long)(p.k1 + kf) * idim + (long)tile * 64;
        const T *u_b = p.u + e3, *u1_b = p.u_1 + e3, *ft_b = p.ft + e3, *ww1_b = p.ww_1 + e3;
        const T *v1n_b = p.v_1 + e3 + js;                                    // row j+1
        const T *t1_b = p.t_1 + e3, *v_b = p.v + e3, *v1_b = p.v_1 + e3;      // row ja (DMA sources advance by rows)
        T *t_b = p.t + e3, *tave_b = p.t_ave + e3, *ww_b = p.ww + e3;

        // DMA lane roles: lane -> (level within the instruction, 16-byte chunk of the 64-element row).
        // Addresses are (wave-uniform base) + (32-bit per-lane offset), like every other access.
        const int dl = lane / LPL;                                           // level within the instruction
        const unsigned dvo = (unsigned)dl * lev + (unsigned)(lane % LPL) * 16u;
        const bool dok = tile * 64 + (lane % LPL) * EPL < p.idim;            // chunk lies inside the memory row
        // copies levels [kf .. kf+KPT) of j row (ja + rows) of `src` (uniform base at row ja) to lds
        auto dma_rows = [&](const T *src, int rows, T *lds) {
#pragma unroll
            for (int q = 0; q < KPT / LPI; ++q) {
                const char *ub = reinterpret_cast<const char *>(src) + (FULL ? (size_t)(q * LPI) * lev : (size_t)0);
                unsigned ro = dvo + (unsigned)rows * row3;                   // the row advance rides in the lane offset
                if (!FULL) {                                                 // virtual levels read the last real one
                    const int l = q * LPI + dl;
                    ro = (unsigned)(l < nlev ? l : nlev - 1) * lev + (unsigned)(lane % LPL) * 16u + (unsigned)rows * row3;
                }
                if (dok)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + ro),
                                                     (__attribute__((address_space(3))) void *)(lds + (size_t)(kf + q * LPI) * 64),
                                                     16, 0, AMT_NT_DMA);
            }
        };
        // i halo of a t_1 row: per level the elements left of lane 0 and right of lane 63, DMA'd one
        // dword per lane into TH[level][side] (2*DPE lanes per instruction, one instruction per level)
        constexpr int DPE = (int)sizeof(T) / 4;                              // dwords per element
        const int hside = lane / DPE;                                        // 0 left, 1 right (lanes < 2*DPE)
        const bool hok = lane < 2 * DPE && (hside == 0 ? tile * 64 - 1 >= 0 : tile * 64 + 64 < p.idim);
        const unsigned hvo = (unsigned)(hside ? 65 * (int)sizeof(T) : 0) + (unsigned)(lane % DPE) * 4u;   // from element -1
        auto dma_halo = [&](const T *src, int rows, T *lds) {
            const unsigned ro = hvo + (unsigned)rows * row3;
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                const char *ub = reinterpret_cast<const char *>(src - 1) + (size_t)lv(m) * lev;
                if (hok)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + ro),
                                                     (__attribute__((address_space(3))) void *)(lds + (size_t)(kf + m) * 2),
                                                     4, 0, 0);
            }
        };

        // element i+64 (right of lane 63) of a u / u_1 row: DPE lanes, one dword each, into UH[which][level]
        const bool uok = lane < DPE && tile * 64 + 64 < p.idim;
        const unsigned uvo = 64u * (unsigned)sizeof(T) + (unsigned)(lane % DPE) * 4u;
        auto dma_uhalo = [&](const T *src, int rows, T *lds) {
            const unsigned ro = uvo + (unsigned)rows * row3;
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                const char *ub = reinterpret_cast<const char *>(src) + (size_t)lv(m) * lev;
                if (uok)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + ro),
                                                     (__attribute__((address_space(3))) void *)(lds + (size_t)(kf + m)),
                                                     4, 0, 0);
            }
        };
        // the DMA set of one row advance: what P1 of row (ja + r) needs beyond what is already in LDS
        auto dma_next = [&](int r, T *t1dst, T *thdst) {
            dma_rows(t1_b, r + 1, t1dst);                                    // t_1(j+1) of that row
            dma_rows(v_b, r + 1, V);                                         // v(j+1)
            dma_halo(t1_b, r + 1, thdst);
            if (XD >= 1) dma_rows(v1_b, r + 1, V1);                          // v_1(j+1)
            if (XD >= 2) { dma_rows(u_b, r, U); dma_uhalo(u_b, r, UH); }     // u(j), u(i+64)
            if (XD >= 3) { dma_rows(u1_b, r, U1); dma_uhalo(u1_b, r, UH + nkr); }
        };

        T vfm[KPT], vft[KPT];
#pragma unroll
        for (int m = 0; m < KPT; ++m) { vfm[m] = vft[m] = T(0); }

        // ---- prologue: everything P1 of row ja needs, by DMA; j-face fluxes of row ja ----
        dma_rows(t1_b, 0, T1);
        dma_halo(t1_b, 0, TH);
        dma_next(0, T1 + t1buf, TH + thbuf);
        {
            T muv_j = T(0), mvx_j = T(0);
            if (act) { muv_j = amt_ld(p.muv + e2, vo); mvx_j = amt_ld(p.msfvx_inv + e2, vo); }
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                const unsigned om = vo + (unsigned)lv(m) * lev;
                if (act) {
                    const T vv = amt_ld(v_b, om);
                    vfm[m] = vv + muv_j * amt_ld(p.v_1 + e3, om) * mvx_j;
                    vft[m] = vv * (amt_ld(t1_b, om) + amt_ld(t1_b - js, om));
                }
            }
        }
        __syncthreads();                               // everything staged, DMA landed

        unsigned o3 = vo;
        for (int jj = ja; jj <= jb; ++jj, o3 += row3) {
            T hf[KPT], tw[KPT];
            T msfty = T(1), mu_tend = T(0), tw_above = T(0);
            const int par = (jj - ja) & 1;
            const T *T1c = T1 + par * t1buf;                         // t_1 row j
            const T *T1n = T1 + (par ^ 1) * t1buf;                   // t_1 row j+1 (DMA'd during the previous row)
            const T *THc = TH + par * thbuf;
            const bool more = (jj < jb);

            // ---------------- P1 ----------------
            if (act) {
                const T msftx = D2[0 * TW + 1 + lane];
                msfty = D2[1 * TW + 1 + lane];
                const T mm = msftx * msfty;
                const T muu_i = D2[2 * TW + 1 + lane], muu_ip = D2[2 * TW + 2 + lane];
                const T msfuy_i = D2[3 * TW + 1 + lane], msfuy_ip = D2[3 * TW + 2 + lane];
                const T muv_p = D2[4 * TW + 1 + lane], mvx_p = D2[5 * TW + 1 + lane];
                mu_tend = D2[6 * TW + 1 + lane];
                if (has_above) {
                    const int Ka = kf + KPT;
                    tw_above = s_fnm[Ka] * T1c[Ka * 64 + lane] + s_fnp[Ka] * T1c[(Ka - 1) * 64 + lane];
                }
                const int ll = lane > 0 ? lane - 1 : 0, lr = lane < 63 ? lane + 1 : 63;
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + (unsigned)lv(m) * lev;
                    const int K = kf + m;
                    T v1n, uu, uup, u1, u1p;
                    if (XD >= 1) v1n = V1[K * 64 + lane]; else v1n = amt_ld(v1n_b, om);
                    if (XD >= 2) {
                        uu = U[K * 64 + lane];
                        const T up_in = U[K * 64 + lr];
                        uup = lane == 63 ? UH[K] : up_in;
                    } else { uu = amt_ld_stream<2>(u_b, om); uup = amt_ld_stream<2>(u_b + 1, om); }
                    if (XD >= 3) {
                        u1 = U1[K * 64 + lane];
                        const T up_in = U1[K * 64 + lr];
                        u1p = lane == 63 ? UH[nkr + K] : up_in;
                    } else { u1 = amt_ld_stream<2>(u1_b, om); u1p = amt_ld_stream<2>(u1_b + 1, om); }
                    const T vn = V[K * 64 + lane], t1n = T1n[K * 64 + lane];
                    const T t1c = T1c[K * 64 + lane];
                    const T t1l_in = T1c[K * 64 + ll], t1r_in = T1c[K * 64 + lr];
                    const T t1l = lane == 0 ? THc[K * 2] : t1l_in;
                    const T t1r = lane == 63 ? THc[K * 2 + 1] : t1r_in;
                    const T vfm_n = vn + muv_p * v1n * mvx_p;
                    const T d = mm * ( rdy * (vfm_n - vfm[m])
                                     + rdx * ( (uup + muu_ip * u1p / msfuy_ip)
                                             - (uu  + muu_i  * u1  / msfuy_i ) ));
                    AB[K * 64 + lane] = d;                        // :142-146; dnw(k)*d is formed by the column wave
                    const T vft_n = vn * (t1n + t1c);
                    hf[m] = msftx * ( hrdy * (vft_n - vft[m])
                                    + hrdx * ( uup * (t1r + t1c) - uu * (t1c + t1l) ) );
                    const T t1km1 = (K > 0) ? T1c[(K > 0 ? K - 1 : 0) * 64 + lane] : T(0);
                    tw[m] = s_fnm[K] * t1c + s_fnp[K] * t1km1;
                    vfm[m] = vfm_n; vft[m] = vft_n;
                }
            }
            __syncthreads();                                         // 1: AP complete; row j of T1/TH/V/D2 dead

            if (more)   // what the next row's P1 needs: no registers, lands before barrier 4
                dma_next(jj - ja + 1, T1 + par * t1buf, TH + par * thbuf);
            T told[KPT], ftk[KPT], w1[KPT];
            T w1_above = T(0);
            if (act) {
                if (has_above) w1_above = amt_ld(ww1_b, o3 + (unsigned)KPT * lev);
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + (unsigned)lv(m) * lev;
                    told[m] = amt_ld_stream<1>(t_b, om);
                    ftk[m] = amt_ld_stream<1>(ft_b, om);
                    w1[m] = amt_ld_stream<1>(ww1_b, om);
                }
            }
            amt_lds_barrier();                                       // 2: DM published (DMA keeps flying)
            T inc_last = T(0);                                      // my top level's increment (:161)
            if (act) {
                const T dmdt = DM[lane];
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const int K = kf + m;
                    const T inc = s_dnw[K] * (dmdt + AB[K * 64 + lane] + mu_tend) / msfty;   // :161
                    AB[K * 64 + lane] = inc;
                    if (m == KPT - 1) inc_last = inc;
                }
            }
            amt_lds_barrier();                                       // 3: AB holds the increments
            __syncthreads();                                         // 4: ww of the recurrence published; DMA landed

            // ---------------- P3 ----------------
            if (act) {
                T wwu = AB[kf * 64 + lane];                          // ww of :161 at my first level
                T wd_k = (kf == 0) ? T(0) : (wwu - w1[0]) * tw[0];
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + (unsigned)lv(m) * lev;
                    const int K = kf + m;
                    const bool real = FULL || m < nlev;                // wave-uniform
                    const T wout = wwu - w1[m];
                    if (real) amt_st_stream(ww_b, om, wout);
                    T wd_n = T(0);
                    const T wwu_n = (m + 1 < KPT) ? AB[(m + 1 < KPT ? K + 1 : K) * 64 + lane] : wwu - inc_last;
                    if (m + 1 < KPT) wd_n = (wwu_n - w1[m + 1 < KPT ? m + 1 : 0]) * tw[m + 1 < KPT ? m + 1 : 0];
                    else if (has_above) wd_n = (wwu_n - w1_above) * tw_above;
                    if (!FULL && K + 1 >= nk) wd_n = T(0);               // wdtn(kde) = 0, :221
                    if (real) amt_st_stream(tave_b, om, told[m]);
                    const T tb = told[m] + msfty * dts * ftk[m];
                    if (real) amt_st_stream(t_b, om, tb - dts * msfty * ( hf[m] + s_rdnw[K] * (wd_n - wd_k) ));
                    wwu = wwu_n; wd_k = wd_n;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------

template <typename T> static size_t amt_march_lds(int nk, int kpt)
{
    const size_t nkr = (size_t)((nk + kpt - 1) / kpt) * kpt;      // nk rounded up to whole cell waves
    // AP [nkr][64]; T1 [2][nkr][66]; D2 [2][7][66]; DM [64]; S1 [4][nkr]
    return ((size_t)nkr * 64 + (size_t)2 * nkr * AMT_TW + 2 * AMT_N2D * AMT_TW + 64 + 4 * nkr) * sizeof(T);
}

template <typename T> static size_t amt_march_dma_lds(int nk, int kpt, int xd)
{
    // AB, T1[2], V (+ V1, U, U1): [nkr][64]; TH [2][nkr][2] (+ UH [2][nkr]); D2 [7][66]; DM [64]; S1 [4][nkr]
    const size_t nkr = (size_t)((nk + kpt - 1) / kpt) * kpt;
    return ((size_t)(4 + xd) * nkr * 64 + (size_t)4 * nkr + (xd >= 2 ? (size_t)2 * nkr : 0)
            + AMT_N2D * AMT_TW + 64 + 4 * nkr) * sizeof(T);
}

// The LDS-DMA flavour moves 16-byte chunks: rows a multiple of 16 bytes, 16-byte aligned bases.
template <typename T> static bool amt_march_dma_layout_ok(const AmtParams<T> &p)
{
    constexpr int EPL = 16 / (int)sizeof(T);
    if (p.idim % EPL != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.t_1) | reinterpret_cast<uintptr_t>(p.v) | reinterpret_cast<uintptr_t>(p.v_1)
         | reinterpret_cast<uintptr_t>(p.u) | reinterpret_cast<uintptr_t>(p.u_1)) & 15u) return false;
    return amt_env_int("AMT_MARCH_DMA", 1) != 0;
}

// Levels per cell wave.  Workgroup size (cell waves + the column wave) is bounded by the
// kernel's __launch_bounds__: 16 waves (KPT < 8), 11 waves (KPT 8..10), 5 waves (KPT 15), and
// the level buffers must fit the 160 KB of LDS.  Preference, measured (profiles/ab_libs.py):
// the smallest KPT the LDS-DMA flavour takes (whole DMA instructions per wave: a multiple of 2 in
// fp64, of 4 in fp32) -- most waves to hide latency, fewest registers per lane; nk need not be a
// multiple of it (virtual levels).  fp64 keeps ~9 values per level live: KPT 5 and 6 spill inside
// the 128-VGPR budget of a 16-wave workgroup, KPT 8 has the 170 of an 11-wave one and beats them
// (NK 64: 5.71 vs 6.11 ms; NK 72: 6.16 vs 6.39 ms per 4096 x NK x 1024 sweep).
template <typename T> static int amt_march_kpt(const AmtParams<T> &p)
{
    const int nk = p.nk;
    if (nk < 1) return 0;
    constexpr int LPI = sizeof(T) == 8 ? 2 : 4;                        // levels per DMA instruction
    static const int pref64[] = {2, 4, 8, 6, 5, 10, 15}, pref32[] = {4, 8, 5, 6, 10, 15, 2};
    const int *pref = sizeof(T) == 8 ? pref64 : pref32;
    auto maxc = [](int k) { return k >= 15 ? 4 : k >= 8 ? 10 : 15; };  // cell waves
    const bool dma_layout = amt_march_dma_layout_ok(p);
    auto feasible = [&](int k) {
        if ((nk + k - 1) / k > maxc(k)) return false;
        if (dma_layout && k % LPI == 0 && amt_march_dma_lds<T>(nk, k, 0) <= 160 * 1024) return true;
        return amt_march_lds<T>(nk, k) <= 160 * 1024;
    };
    const int forced = amt_env_int("AMT_MARCH_KPT", 0);
    for (int i = 0; i < 7; ++i)
        if (pref[i] == forced && feasible(forced)) return forced;
    for (int i = 0; i < 7; ++i)
        if (feasible(pref[i])) return pref[i];
    return 0;
}

template <typename T> static long amt_march_max_rows(const AmtParams<T> &p)
{
    // the per-lane byte offsets of the march are 32-bit
    const long row_bytes = p.jstride * (long)sizeof(T);
    return ((1L << 31) - 16L * p.idim * (long)sizeof(T)) / row_bytes - 3;
}

template <typename T> bool amt_march_supported(const AmtParams<T> &p)
{
    return amt_march_kpt<T>(p) != 0 && amt_march_max_rows(p) >= 1;
}

template <typename T, int KPT, bool FULL>
static hipError_t amt_march_launch_full(hipStream_t stream, const AmtParams<T> &p, const AmtMarchGrid &g, size_t lds)
{
    const int nw = (p.nk + KPT - 1) / KPT + 1;       // cell waves + the column wave
    if (lds > 64 * 1024) {
        // the attribute is per device and per kernel instantiation; remember what was granted
        static thread_local size_t granted[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int slot = (dev >= 0 && dev < 64) ? dev : 0;
        if (lds > granted[slot] || slot != dev) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(amt_march_kernel<T, KPT, FULL, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            granted[slot] = lds;
        }
    }
    if (KPT == 4 && FULL && sizeof(T) == 8 && amt_env_int("AMT_MARCH_STAMP", 0)) {
        // diagnostic instantiation: per-phase cycle sums of every wave, printed to stderr
        // (its fences forbid overlaps the real kernel has: read shares, never quote its run time)
        AmtMarchGrid gs = g;
        const size_t n = (size_t)g.nwg * 16 * 8;
        if (hipMalloc((void **)&gs.stamps, n * 8) != hipSuccess) return hipGetLastError();
        (void)hipMemsetAsync(gs.stamps, 0, n * 8, stream);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(amt_march_kernel<T, 4, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((amt_march_kernel<T, 4, true, true>), dim3(g.nwg), dim3(nw * 64), lds, stream, p, gs);
        (void)hipStreamSynchronize(stream);
        unsigned long long *h = (unsigned long long *)malloc(n * 8);
        (void)hipMemcpy(h, gs.stamps, n * 8, hipMemcpyDeviceToHost);
        double cell[8] = {0}, col[8] = {0}, w0[8] = {0};
        for (size_t b = 0; b < (size_t)g.nwg; ++b)
            for (int ww = 0; ww < nw; ++ww)
                for (int q = 0; q < 8; ++q) {
                    const double x = (double)h[(b * 16 + ww) * 8 + q];
                    if (ww == nw - 1) col[q] += x; else cell[q] += x;
                    if (ww == 0) w0[q] += x;
                }
        const char *names[8] = {"P1", "bar1", "P2a", "bar2", "P2b", "bar3", "idle", "bar4+P3"};
        const double rows = (double)g.nwg * g.jrows;
        fprintf(stderr, "[amt stamps] cycles per row: phase cell-mean / wave0 / column-wave\n[amt stamps]");
        for (int q = 0; q < 8; ++q)
            fprintf(stderr, "  %s %.0f/%.0f/%.0f", names[q], cell[q] / (rows * (nw - 1)), w0[q] / rows, col[q] / rows);
        fprintf(stderr, "\n");
        free(h);
        (void)hipFree(gs.stamps);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((amt_march_kernel<T, KPT, FULL, false>), dim3(g.nwg), dim3(nw * 64), lds, stream, p, g);
    return hipGetLastError();
}

template <typename T, int KPT>
static bool amt_march_dma_ok(const AmtParams<T> &p)
{
    constexpr int EPL = 16 / (int)sizeof(T), LPI = 64 / (64 / EPL);
    return KPT % LPI == 0 && amt_march_dma_layout_ok(p) && amt_march_dma_lds<T>(p.nk, KPT, 0) <= 160 * 1024;
}

template <typename T, int KPT, int XD, bool FULL>
static hipError_t amt_march_launch_dma_xd(hipStream_t stream, const AmtParams<T> &p, const AmtMarchGrid &g)
{
    const size_t lds = amt_march_dma_lds<T>(p.nk, KPT, XD);
    const int nw = (p.nk + KPT - 1) / KPT + 1;
    if (lds > 64 * 1024) {
        static thread_local size_t granted[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int slot = (dev >= 0 && dev < 64) ? dev : 0;
        if (lds > granted[slot] || slot != dev) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(amt_march_dma_kernel<T, KPT, XD, FULL>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            granted[slot] = lds;
        }
    }
    hipLaunchKernelGGL((amt_march_dma_kernel<T, KPT, XD, FULL>), dim3(g.nwg), dim3(nw * 64), lds, stream, p, g);
    return hipGetLastError();
}

template <typename T, int KPT>
static hipError_t amt_march_launch_dma(hipStream_t stream, const AmtParams<T> &p, const AmtMarchGrid &g)
{
    if constexpr (KPT % (64 / (64 / (16 / (int)sizeof(T)))) == 0) {
        // more inputs through the DMA where measured to pay (in-process A/B, profiles/ab_libs.py):
        // fp32: all of them (4096x60x4096: 8.27 vs 8.75 ms; NK 80: 11.4 vs 13.2 ms), never u without
        // u_1 (XD 2 is the slowest everywhere); fp64: at KPT 4 the u/u_1 reads from LDS cost ~10 more
        // live VGPRs and spill (NK 40: 11.86 vs 11.26 ms; NK 60: v_1 alone 15.77 vs 15.72 without), at
        // KPT 2 (NK <= 30) they fit and pay (NK 20: 5.62 vs 5.82 ms)
        int xd = 3;
        while (xd > 0 && amt_march_dma_lds<T>(p.nk, KPT, xd) > 160 * 1024) --xd;
        if (xd == 2) xd = 1;
        if (sizeof(T) == 8 && !(xd == 3 && KPT <= 2)) xd = 0;
        const int cap = amt_env_int("AMT_MARCH_XD", -1);
        if (cap >= 0) {
            xd = cap > 3 ? 3 : cap;
            while (xd > 0 && amt_march_dma_lds<T>(p.nk, KPT, xd) > 160 * 1024) --xd;
        }
        const bool full = p.nk % KPT == 0;
        switch (xd) {
        case 3:  return full ? amt_march_launch_dma_xd<T, KPT, 3, true>(stream, p, g) : amt_march_launch_dma_xd<T, KPT, 3, false>(stream, p, g);
        case 2:  return full ? amt_march_launch_dma_xd<T, KPT, 2, true>(stream, p, g) : amt_march_launch_dma_xd<T, KPT, 2, false>(stream, p, g);
        case 1:  return full ? amt_march_launch_dma_xd<T, KPT, 1, true>(stream, p, g) : amt_march_launch_dma_xd<T, KPT, 1, false>(stream, p, g);
        default: return full ? amt_march_launch_dma_xd<T, KPT, 0, true>(stream, p, g) : amt_march_launch_dma_xd<T, KPT, 0, false>(stream, p, g);
        }
    } else {
        return hipErrorNotSupported;
    }
}

template <typename T, int KPT>
static hipError_t amt_march_launch_kpt(hipStream_t stream, const AmtParams<T> &p, const AmtMarchGrid &g, size_t lds)
{
    if (amt_march_dma_ok<T, KPT>(p) && !amt_env_int("AMT_MARCH_STAMP", 0))
        return amt_march_launch_dma<T, KPT>(stream, p, g);
    return (p.nk % KPT == 0) ? amt_march_launch_full<T, KPT, true>(stream, p, g, lds)
                             : amt_march_launch_full<T, KPT, false>(stream, p, g, lds);
}

template <typename T>
hipError_t amt_launch_march(hipStream_t stream, const AmtParams<T> &p)
{
    const int ni = p.i1 - p.i0 + 1, nj = p.j1 - p.j0 + 1;
    if (ni <= 0 || nj <= 0) return hipSuccess;
    const int kpt = amt_march_kpt<T>(p);
    if (kpt == 0) return hipErrorNotSupported;
    AmtMarchGrid g;
    g.stamps = nullptr;
    g.tile_lo = p.i0 / 64;
    g.ntile_i = p.i1 / 64 - g.tile_lo + 1;
    // Rows per workgroup.  A block costs its rows plus a prologue (4 extra array-rows of loads,
    // about half a row of time); workgroups run in rounds of `slots` = CUs x resident
    // workgroups per CU, so the sweep takes about  rounds(r) * (r + 0.5)  row-times.  Pick the r
    // that minimises it: for the whole 4096-row domain any r near 32 is within 1 %, but for a
    // 510-row j-slab (8 GPUs) r = 32 would leave the last round 6 % full (1040 workgroups on 256
    // CUs) and cost 17 % more than r = 11.  Small launches get short blocks, down to one row (more
    // workgroups: 64x40x64 takes 12 us with r = 1, 27 us with r = 4).
    int jrows = amt_env_int("AMT_MARCH_JROWS", 0);
    if (jrows < 1) {
        static int slots = 0;
        if (slots == 0) {
            int dev = 0, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            (void)hipGetLastError();
            slots = cus;
        }
        const size_t lds_need = amt_march_lds<T>(p.nk, kpt);
        const int per_cu = (int)((160u * 1024u) / (lds_need ? lds_need : 1));     // LDS is what bounds residency
        const long sl = (long)slots * (per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu);
        double best = 1e300;
        for (int r = 1; r <= 64 && r <= nj; ++r) {
            const long blocks = (long)g.ntile_i * ((nj + r - 1) / r);
            const long rounds = (blocks + sl - 1) / sl;
            const double cost = (double)rounds * (r + 0.5);
            if (cost < best - 1e-9 || (cost < best + 1e-9 && r > jrows)) { best = cost; jrows = r; }
        }
    }
    if (jrows > nj) jrows = nj;
    const long max_rows = amt_march_max_rows(p);
    if (max_rows < 1) return hipErrorNotSupported;
    if (jrows > max_rows) jrows = (int)max_rows;
    g.jrows = jrows;
    g.njblk = (nj + jrows - 1) / jrows;
    g.nwg = g.ntile_i * g.njblk;
    const size_t lds = amt_march_lds<T>(p.nk, kpt);
    switch (kpt) {
    case 2:  return amt_march_launch_kpt<T, 2>(stream, p, g, lds);
    case 4:  return amt_march_launch_kpt<T, 4>(stream, p, g, lds);
    case 5:  return amt_march_launch_kpt<T, 5>(stream, p, g, lds);
    case 6:  return amt_march_launch_kpt<T, 6>(stream, p, g, lds);
    case 8:  return amt_march_launch_kpt<T, 8>(stream, p, g, lds);
    case 10: return amt_march_launch_kpt<T, 10>(stream, p, g, lds);
    case 15: return amt_march_launch_kpt<T, 15>(stream, p, g, lds);
    default: return hipErrorNotSupported;
    }
}

template bool amt_march_supported<float>(const AmtParams<float> &);
template bool amt_march_supported<double>(const AmtParams<double> &);
template hipError_t amt_launch_march<float>(hipStream_t, const AmtParams<float> &);
template hipError_t amt_launch_march<double>(hipStream_t, const AmtParams<double> &);
