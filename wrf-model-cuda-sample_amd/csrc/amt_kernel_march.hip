// amt_kernel_march.hip -- AMT_VARIANT_MARCH: (i,k)-cell lanes marching in j.
//
// Why: the column kernel reads v, v_1 twice (rows j and j+1) and t_1 three times
// (j-1, j, j+1) from beyond L2 -- rocprofv3 shows 1.97x the compulsory read bytes
// (profiles/r01_column_pmc.json).  This kernel reads every input element once per sweep:
//
//  * A workgroup owns one i-tile of TC columns (memory aligned), ALL levels, and a block of
//    consecutive j rows that it marches through.  Every 3-D access of a wave is made of
//    contiguous runs along i (i-contiguous coalesced loads).
//  * Shape of a wave (template parameters; the launcher picks them per level count, precision
//    and layout -- see amt_march_plan below):
//      VW  columns per lane (1, or 2: fp32 with 8-byte accesses per lane -- the fp64 profile);
//      HL  level groups per wave: the 64 lanes are HL groups of TI = 64/HL lanes, group h owns
//          levels [h*KPT, (h+1)*KPT) of the wave's LW = HL*KPT levels, so a tile is TC = TI*VW
//          columns wide.  HL = 2, 4 halve/quarter every [level][column] LDS buffer and the
//          registers a level count needs: fp64 runs 61..120 levels with the 4 levels per lane
//          that 60 levels use (no scratch), up to 240 (fp32: 264) with HL = 4, where LDS ends it;
//      KPT levels per lane.
//    NC "cell" waves own LW consecutive levels each; one more "column" wave owns everything
//    that is per column (i,j) rather than per cell: the two sequential k chains, the 2-D mass
//    update, and the staging of the 2-D input rows (wave specialisation).
//  * The j-direction face fluxes  v(j)+muv(j)*v_1(j)*msfvx_inv(j)  and
//    v(j)*(t_1(j)+t_1(j-1))  are carried in registers from one row to the next (the face
//    j+1 of row j IS the face j of row j+1: the same expression on the same operands,
//    so carrying it is bit-exact); the t_1 row is carried in LDS (with its i halo), which
//    also serves t_1(i-1), t_1(i+1) and t_1(k-1).  v, v_1, t_1 are read once.
//  * The k-dependencies go through LDS (k-column staging), four barriers per j row:
//      cell waves   AB[k][col] = dvdxi(i,k)                                 (:142-146)
//      -- barrier 1 --
//      column wave  dmdt = sum_k dnw(k)*AB[k] in the Fortran's sequential k order -> DM[col]
//      -- barrier 2 --
//      cell waves   AB[k][col] <- dnw(k)*(dmdt+dvdxi(k)+mu_tend)/msfty  (:161, one divide
//                   per cell, own slots)
//      -- barrier 3 --
//      column wave  AB[k] <- ww(k); ww(k+1) = ww(k) - AB[k], sequential  (:161)
//      cell waves   store t_ave = t as it came in (:211): the one store that waits for no chain
//      -- barrier 4 --
//      cell waves   ww - ww_1 (:170), wdtn (:220-227), theta update (:212, :237-246)
//      column wave  the 2-D mass update (:151-157) -- behind the chain, not in front of it: its
//                   stores' round trip to memory must not sit on the row's critical path
//    A column's chains are summed exactly once, in order: bit-exact and no redundant LDS
//    traffic (an earlier version let every wave redo both chains: LDS-bandwidth bound).
//  * Level counts that do not fill the cell waves: the missing levels of the last wave are
//    virtual (clamped loads of the last real level, no stores) -- no per-level branches.
//  * Expressions keep the Fortran association; built with -ffp-contract=off.
//  * Two flavours (template DMA): with DMA the rows j+2 of t_1 and v (and, where LDS allows,
//    v_1, u, u_1: template XD) that the NEXT row's P1 needs are fetched by LDS-DMA
//    (`global_load_lds_dwordx4`, no VGPR destination) right after barrier 1, i.e. while the
//    column wave runs its chains and the cell waves would otherwise only wait.  Barriers 2 and 3
//    are LDS-only (inline asm): a __syncthreads() would drain the DMA (hipcc waits vmcnt(0) at a
//    workgroup fence while an LDS-DMA is in flight); barrier 4 is a full one and is where the DMA
//    must have landed.  The DMA's global source needs no alignment (profiles/r02_lds_dma_alignment.md),
//    so every layout takes this flavour; with DMA = false (AMT_MARCH_DMA=0, kept as a cross-check) the
//    same rows go through registers inside P1.
//
// Reference semantics: module_small_step_em.f90:112-172 (mu, ww), :208-215 and
// :217-250 (theta); the fusion of the three Fortran phases is legal because a
// column never reads another column's outputs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>
#include <type_traits>
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
#ifndef AMT_TAVE_EARLY
#define AMT_TAVE_EARLY 1 /* t_ave = the row's incoming t (:211) is stored between barriers 3 and 4, while the column wave runs
                            its second chain and the cell waves would only wait: the one store of P3 that depends on neither chain */
#endif
#ifndef AMT_P1_BATCH
#define AMT_P1_BATCH -1 /* levels whose global loads P1 issues together before it consumes any: -1 the instantiation's own choice
                           (amt_p1_batch), 1 level by level (what hipcc schedules on its own: it keeps register pressure low
                           and pays one full memory latency per LEVEL), KPT all levels of the lane at once */
#endif
#ifndef AMT_CHAIN
#define AMT_CHAIN 10    /* LDS reads kept in flight by the sequential k chains */
#endif
struct AmtMarchGrid {
    int ntile_i;     // number of i tiles that hold window columns
    int col_lo;      // memory column of the first tile's column 0: the window's first column rounded down to a 128-B line
    int jrows;       // rows per workgroup
    int jstep;       // rows from one j block's first row to the next one's (jrows; edge launches: j1 - j0)
    int njblk;       // number of j blocks
    int nwg;         // ntile_i * njblk
    int xchunk;      // consecutive logical ids an XCD takes per round of the launch (0: one run for the whole launch)
};

// ---------------------------------------------------------------------------
// VW columns per lane: a tiny elementwise vector.  Every operator is elementwise and keeps the
// association written at the call site, so a VW = 2 lane computes exactly what two VW = 1 lanes do.
// ---------------------------------------------------------------------------
template <typename T, int VW> struct AmtVec {
    T x[VW];
    __device__ __forceinline__ AmtVec() {}
    __device__ __forceinline__ explicit AmtVec(T s) {
#pragma unroll
        for (int e = 0; e < VW; ++e) x[e] = s;
    }
};
#define AMT_VEC_OP(op)                                                                               \
    template <typename T, int VW> __device__ __forceinline__ AmtVec<T, VW> operator op(const AmtVec<T, VW> &a, const AmtVec<T, VW> &b) \
    { AmtVec<T, VW> r; _Pragma("unroll") for (int e = 0; e < VW; ++e) r.x[e] = a.x[e] op b.x[e]; return r; }          \
    template <typename T, int VW> __device__ __forceinline__ AmtVec<T, VW> operator op(T a, const AmtVec<T, VW> &b)   \
    { AmtVec<T, VW> r; _Pragma("unroll") for (int e = 0; e < VW; ++e) r.x[e] = a op b.x[e]; return r; }               \
    template <typename T, int VW> __device__ __forceinline__ AmtVec<T, VW> operator op(const AmtVec<T, VW> &a, T b)   \
    { AmtVec<T, VW> r; _Pragma("unroll") for (int e = 0; e < VW; ++e) r.x[e] = a.x[e] op b; return r; }
AMT_VEC_OP(+)
AMT_VEC_OP(-)
AMT_VEC_OP(*)
AMT_VEC_OP(/)
#undef AMT_VEC_OP

// memory image of VW consecutive elements; only element alignment is promised (a WRF row may
// start anywhere), gfx950 global memory takes misaligned dwordx2 accesses
template <typename T, int VW> struct AmtRaw;
template <typename T> struct AmtRaw<T, 1> { typedef T type; };
template <> struct AmtRaw<float, 2> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct AmtRaw<double, 2> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };

template <typename T, int VW> __device__ __forceinline__ AmtVec<T, VW> amt_unraw(typename AmtRaw<T, VW>::type r)
{
    AmtVec<T, VW> v;
    if constexpr (VW == 1) v.x[0] = r;
    else {
#pragma unroll
        for (int e = 0; e < VW; ++e) v.x[e] = r[e];
    }
    return v;
}
template <typename T, int VW> __device__ __forceinline__ typename AmtRaw<T, VW>::type amt_raw(const AmtVec<T, VW> &v)
{
    if constexpr (VW == 1) return v.x[0];
    else {
        typename AmtRaw<T, VW>::type r;
#pragma unroll
        for (int e = 0; e < VW; ++e) r[e] = v.x[e];
        return r;
    }
}

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
template <typename T, int VW>
__device__ __forceinline__ AmtVec<T, VW> amt_ldv(const T *ubase, unsigned voff)
{
    typedef typename AmtRaw<T, VW>::type R;
    return amt_unraw<T, VW>(*reinterpret_cast<const R *>(reinterpret_cast<const char *>(ubase) + voff));
}
// once-read inputs (LEVEL = which setting of the kernel's NTL turns the nt policy on for this stream)
template <int LEVEL, int NTL, typename T, int VW>
__device__ __forceinline__ AmtVec<T, VW> amt_ldv_stream(const T *ubase, unsigned voff)
{
    typedef typename AmtRaw<T, VW>::type R;
    if constexpr (NTL >= LEVEL)
        return amt_unraw<T, VW>(__builtin_nontemporal_load(reinterpret_cast<const R *>(reinterpret_cast<const char *>(ubase) + voff)));
    else
        return amt_unraw<T, VW>(*reinterpret_cast<const R *>(reinterpret_cast<const char *>(ubase) + voff));
}
template <int LEVEL, int NTL, typename T>
__device__ __forceinline__ T amt_ld_stream(const T *ubase, unsigned voff)
{
    if constexpr (NTL >= LEVEL)
        return __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(ubase) + voff));
    else
        return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(ubase) + voff);
}
// store of a lane's VW columns: one wide store when all of them are window columns (every lane
// but the one or two the window edge cuts), element stores otherwise
template <typename T, int VW>
__device__ __forceinline__ void amt_stv(T *ubase, unsigned voff, const AmtVec<T, VW> &v, bool all, const bool (&on)[VW])
{
    typedef typename AmtRaw<T, VW>::type R;
    if constexpr (VW == 1) {
        if (on[0]) {
#if AMT_NT_STORE
            __builtin_nontemporal_store(v.x[0], reinterpret_cast<T *>(reinterpret_cast<char *>(ubase) + voff));
#else
            *reinterpret_cast<T *>(reinterpret_cast<char *>(ubase) + voff) = v.x[0];
#endif
        }
    } else {
        if (all) {
            *reinterpret_cast<R *>(reinterpret_cast<char *>(ubase) + voff) = amt_raw<T, VW>(v);
        } else {
#pragma unroll
            for (int e = 0; e < VW; ++e)
                if (on[e]) *reinterpret_cast<T *>(reinterpret_cast<char *>(ubase) + voff + e * (unsigned)sizeof(T)) = v.x[e];
        }
    }
}
template <typename T, int VW> __device__ __forceinline__ AmtVec<T, VW> amt_ldsv(const T *q)
{
    AmtVec<T, VW> v;
#pragma unroll
    for (int e = 0; e < VW; ++e) v.x[e] = q[e];
    return v;
}
template <typename T, int VW> __device__ __forceinline__ void amt_stsv(T *q, const AmtVec<T, VW> &v)
{
#pragma unroll
    for (int e = 0; e < VW; ++e) q[e] = v.x[e];
}

#ifndef AMT_STAMPS
#define AMT_STAMPS 0    /* 1: instrumentation build (profiles/stamps.py): the middle workgroup records s_memtime at the phase
                           boundaries of rows 4..35 of its block -- first cell wave, last cell wave, column wave */
#endif
#if AMT_STAMPS
__device__ unsigned long long amt_stamp_buf[3][32][8];
#define AMT_STAMP(slot, n) do { if (stamp_on && jj - ja >= 4 && jj - ja < 36) { unsigned long long t_;                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                                  \
        if (lane == 0) amt_stamp_buf[slot][jj - ja - 4][n] = t_; } } while (0)
extern "C" int amt_diag_stamps(void *out, int bytes)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(amt_stamp_buf), (size_t)bytes < sizeof amt_stamp_buf ? (size_t)bytes : sizeof amt_stamp_buf);
}
// every workgroup: REFCLK (100 MHz, one clock for the whole chip) when its first wave starts and when that wave leaves its last row,
// and where it ran (HW_ID, XCC_ID) -- profiles/spans.py
__device__ unsigned long long amt_span_buf[16384][4];
#define AMT_SPAN(n) do { if (threadIdx.x == 0 && lid < 16384) { unsigned long long t_;                                 \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                              \
        amt_span_buf[lid][n] = t_;                                                                                     \
        if (n == 0) { unsigned h_, x_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h_));                \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_));                                          \
            amt_span_buf[lid][2] = h_; amt_span_buf[lid][3] = x_; } } } while (0)
extern "C" int amt_diag_spans(void *out, int bytes)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(amt_span_buf), (size_t)bytes < sizeof amt_span_buf ? (size_t)bytes : sizeof amt_span_buf);
}
#else
#define AMT_STAMP(slot, n) do { } while (0)
#define AMT_SPAN(n) do { } while (0)
#endif

constexpr int AMT_N2D = 7;    // staged 2-D rows: msftx msfty muu msfuy muv' msfvx_inv' mu_tend

__device__ __forceinline__ void amt_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
// XD (DMA flavour): how many MORE input rows ride the DMA when LDS allows:
//   0: t_1, v   1: + v_1 (row j+2)   2: + u (row j+1, with its i+1 halo)   3: + u_1  -> no global load left in P1
// FULL: nk is a multiple of LW.  Otherwise the last cell wave owns fewer real levels; its missing
// ones are VIRTUAL: they load the wave's last real level again (addresses clamped), compute on the
// duplicate and are never stored, so that the code stays free of per-level branches (which would
// break the load batching) -- the LDS level buffers hold nkr = waves*LW rows either way.
// WM: most waves a workgroup of this instantiation is launched with: 16 (4 per SIMD: 128 VGPRs per
// lane) or 12 (3 per SIMD: 168 VGPRs -- what the shapes with level groups need to stay out of scratch).
// Levels per load batch of P1 (see PB in the kernel).
template <typename T, int VW, int KPT, int HL, int XD, bool DMA, int WM> constexpr int amt_p1_batch()
{
    if (AMT_P1_BATCH >= 0) return AMT_P1_BATCH > KPT ? KPT : AMT_P1_BATCH;
    // As many levels as the instantiation's registers hold without scratch (a scratch reload drains the LDS-DMA like any
    // load; `make check` fails the build on scratch in any selectable instantiation).  Measured, same process, same arrays
    // (profiles/r04_raw/ab_p1_batch.txt): all three levels of the (.,3,.) shapes at once -1.4 % (fp32 8192x80x2048), -1.8 %
    // (fp64 4096x80x2048), -0.7 / -1.6 % (40 levels fp64 / fp32); two of the four of <float,2,4,1> -0.4 % (fp32 4096x60x4096);
    // the fp64 (1,4,.,16) shapes -- the headline among them -- and six levels per lane have no registers for it (a batch
    // of two costs the headline 16-24 B of scratch and 0.8 %).
    if (!DMA) return 1;
    if (KPT <= 3) return KPT;
    if (KPT == 4 && WM == 12) return 4;
    if (KPT == 4 && sizeof(T) == 4 && VW == 2 && HL == 1) return 2;
    return 1;
}

// NTL: the cache policy of the once-read streams (0: none; 1: t, ft, ww_1 non-temporal; 2: u, u_1 too).  Same bits, same
// everything else; the launcher instantiates and picks between two:
//   NTL = AMT_NT_LOAD (1)   rows that are whole 128-byte lines: no line of those streams is ever wanted by a second workgroup, and
//                           nt keeps them from evicting the lines that are (-1.4 % of a sweep, -0.8 % traffic on the padded
//                           4096 x 60 x 4096 fp64 state).
//   NTL = 0                 rows that are NOT whole lines (WRF's own ims:ime = 0:NI+1: 4098 x 8 B = 256 lines + 16 B): seven of
//                           eight level rows of a tile end inside a line whose rest is the NEIGHBOUR tile's first columns of the
//                           same streams, and nt asks L2 to drop exactly that line -- HBM reads 70.7 -> 67.7 GB per sweep (1.065
//                           -> 1.031 x the algorithmic bytes; profiles/r06_rows4098_nt.md).
// The launcher picks by (idim * sizeof(T)) % 128 (amt_march_set_stream_policy / AMT_MARCH_NT override it).  (A shared __device__
// body under two __global__ names was tried first: it is not register-neutral -- the <float,2,6,.,12> shapes went from 164 VGPRs
// and no scratch to 167 and 12 B/lane -- a template parameter of the kernel itself is.)
template <typename T, int VW, int KPT, int HL, int XD, bool FULL, bool DMA, int WM, int NTL>
__global__ __launch_bounds__(WM * 64) void amt_march_kernel(const AmtParams<T> p, const AmtMarchGrid g)
{
    extern __shared__ __align__(16) unsigned char amt_smem[];
    typedef AmtVec<T, VW> V;
    constexpr int TI = 64 / HL;                   // lanes per level group
    constexpr int TC = TI * VW;                   // columns per tile
    constexpr int LW = KPT * HL;                  // levels per cell wave
    constexpr int TW = TC + 2;                    // a staged 2-D row: TC columns, the right halo, pad
    constexpr int N2D = AMT_N2D;
    constexpr unsigned W = (unsigned)sizeof(T);
    static_assert(XD == 0 || DMA, "XD counts extra DMA'd inputs");
    const int nk = p.nk;

    const int lane = threadIdx.x & 63;
    const int w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
    const int nwav = (int)(blockDim.x >> 6);
    const int nc   = nwav - 1;                                           // cell waves 0..nc-1
    const bool colw = (w == nc);                                         // the column wave
    const int nkr  = nc * LW;                                            // level rows of the LDS buffers

    // AB[k][col] goes through three lives per row: dvdxi(i,k) (P1 .. barrier 2), the ww increment
    // of level k (barrier 2 .. 3), ww(k) of the recurrence :161 (barrier 4 .. P3).  A cell lane
    // only ever reads its OWN slots after barrier 2, so the next row's P1 may overwrite them without
    // another barrier; the value a lane needs from above its last level is rebuilt from its own last
    // increment, kept in a register (the same subtraction the column wave performs).
    T *AB = reinterpret_cast<T *>(amt_smem);                  // [nkr][TC]
    T *T1 = AB + (size_t)nkr * TC;                            // [2][nkr][TC] t_1 rows (buffer = row parity)
    T *TH = T1 + (size_t)2 * nkr * TC;                        // [2][nkr][2] i halo of the t_1 rows: left, right
    T *VB = TH + (size_t)4 * nkr;                             // [nkr][TC]   v of row j+1            (DMA)
    T *V1 = VB + (DMA ? (size_t)nkr * TC : 0);                // [nkr][TC]   v_1 of row j+1          (XD >= 1)
    T *U  = V1 + (XD >= 1 ? (size_t)nkr * TC : 0);            // [nkr][TC]   u of row j              (XD >= 2)
    T *U1 = U + (XD >= 2 ? (size_t)nkr * TC : 0);             // [nkr][TC]   u_1 of row j            (XD >= 3)
    T *UH = U1 + (XD >= 3 ? (size_t)nkr * TC : 0);            // [2][nkr]    element i+TC of the u / u_1 rows (DMA)
    T *D2 = UH + (DMA ? (size_t)2 * nkr : 0);                 // [N2D][TW]   2-D inputs of the current row
    T *DM = D2 + (size_t)N2D * TW;                            // [3][TC]     of the row: dmdt | mu_tend | msfty
    T *S1 = DM + 3 * TC;                                      // [nkr][4]    dnw, fnm, fnp, rdnw of the level
    T *TWB = S1 + (size_t)4 * nkr;                            // [2][nc*HL][TC] fnm*t_1(k)+fnp*t_1(k-1) at each level group's FIRST level (row parity)
    auto s_dnw = [&](int k) { return S1[4 * k]; };            // (fnm, fnp, rdnw: S1[4k + 1 .. 3])
    const int t1buf = nkr * TC, thbuf = nkr * 2;

    // XCD-aware logical workgroup id: blocks b, b+8, b+16 ... share an XCD (round-robin
    // dispatch), so give each XCD a contiguous run of logical ids: neighbouring i-tiles
    // of one j block then run on one XCD at about the same time and share the tile-edge
    // cache lines in that XCD's L2.  Speed only, never correctness.
    const int nx = 8;
    const int x = blockIdx.x % nx, y = blockIdx.x / nx;
    const int xc = g.xchunk, whole = xc > 0 ? g.nwg / (nx * xc) : 0;          // whole rounds of nx * xc workgroups
    int lid;
    if (y < whole * xc) {
        lid = (y / xc) * (nx * xc) + x * xc + y % xc;                         // round, then XCD x's run of xc ids in it
    } else {
        const int base = whole * nx * xc, rest = g.nwg - base, q = rest / nx, r = rest % nx;
        lid = base + x * q + (x < r ? x : r) + (y - whole * xc);              // XCD x owns q (+1 if x < r) consecutive ids
    }
    const int ja = p.j0 + (lid / g.ntile_i) * g.jstep;
    const int jb = (ja + g.jrows - 1 < p.j1) ? ja + g.jrows - 1 : p.j1;
    // Tiles are anchored at the WINDOW (its first column rounded down to a 128-byte line of the row), not
    // at multiples of TC from the row start: a window that starts 32 elements into a row (the resident
    // layout) would otherwise straddle one more, half-empty tile (512x60x512: 9 x 27 = 243 workgroups on
    // 256 CUs instead of 8 x 32).  Neither the LDS-DMA nor the plain loads need more than line alignment.
    const int col0 = g.col_lo + (lid % g.ntile_i) * TC;
    AMT_SPAN(0);
#if AMT_STAMPS
    const bool stamp_on = (lid == g.nwg / 2 + g.ntile_i / 2) && (w == 0 || w >= nc - 1);
    const int stamp_slot = w == 0 ? 0 : colw ? 2 : 1;
#endif

    for (int e = threadIdx.x; e < 4 * nkr; e += blockDim.x) {
        const int which = e & 3, k = e >> 2;
        const T *src = which == 0 ? p.dnw : which == 1 ? p.fnm : which == 2 ? p.fnp : p.rdnw;
        S1[e] = src[p.k1 + (k < nk ? k : nk - 1)];
    }

    const long idim = p.idim, js = p.jstride;
    const unsigned lev = (unsigned)idim * W;                  // byte step of one level
    const unsigned row3 = (unsigned)js * W;                   // byte step of one j row (3-D)
    const unsigned row2 = lev;                                // byte step of one j row (2-D)
    const long e2 = (long)ja * idim + (long)col0;        // (column 0 of the tile, row ja)
    const bool halo_l_mem = col0 - 1 >= 0;               // the tile's left / right neighbour column exists
    const bool halo_r_mem = col0 + TC < p.idim;

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
        // column wave: lanes 0..TI-1 own the tile's columns (VW each)
        // =====================================================================
        const int c = lane * VW;
        const int ii = col0 + c;
        const bool own = lane < TI;
        const bool inmem = own && (ii + VW <= p.idim);
        const bool tail = VW > 1 && own && !inmem && ii < p.idim;  // odd row length: my first column is the row's last
        bool on[VW], any = false, all = true;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            on[e] = own && (ii + e >= p.i0) && (ii + e <= p.i1);
            any = any || on[e]; all = all && on[e];
        }
        const bool halo_r = (lane == 0) && halo_r_mem;         // lane 0 also fetches element TC
        const unsigned vo = (unsigned)c * W;
        const T *wwin_b = p.ww + (long)ja * js + (long)p.k1 * idim + (long)col0;   // level 1
        const T *mut_b = p.mut + e2;
        T *mu_b = p.mu + e2, *mudf_b = p.mudf + e2, *muts_b = p.muts + e2, *muave_b = p.muave + e2;
        const T dts = p.dts;

        // prologue: 2-D row ja into D2
#pragma unroll
        for (int q = 0; q < N2D; ++q) {
            const T *src = d2_src(q);
            if (inmem) amt_stsv<T, VW>(D2 + q * TW + c, amt_ldv<T, VW>(src, vo));
            if (tail) D2[q * TW + c] = amt_ld(src, vo);
            if (halo_r) D2[q * TW + TC] = amt_ld(src + TC, 0u);
        }
        __syncthreads();                           // S1, T1, TH, (VB ..), D2 staged

        unsigned o3 = vo, o2 = vo;
        for (int jj = ja; jj <= jb; ++jj, o3 += row3, o2 += row2) {
            const bool more = (jj < jb);
            AMT_STAMP(2, 0);
            // while the cell waves do P1: fetch the next 2-D row and this row's column inputs
            V d2v[N2D];
            T d2h[N2D];                                          // lane 0: the right halo; a tail lane: its one column
#pragma unroll
            for (int q = 0; q < N2D; ++q) { d2v[q] = V(T(0)); d2h[q] = T(0); }
            if (more) {
#pragma unroll
                for (int q = 0; q < N2D; ++q) {
                    const T *src = d2_src(q);
                    if (inmem) d2v[q] = amt_ldv<T, VW>(src, o2 + row2);
                    if (halo_r) d2h[q] = amt_ld(src + TC, o2 - vo + row2);
                    if (tail) d2h[q] = amt_ld(src, o2 + row2);
                }
            }
            V ww1in(T(0)), mu_old(T(0)), mut_v(T(0)), mu_tend(T(0)), msfty_c(T(1));
            if (any) {
                ww1in = amt_ldv<T, VW>(wwin_b, o3);                // incoming ww(i,1,j)
                mu_old = amt_ldv<T, VW>(mu_b, o2);
                mut_v = amt_ldv<T, VW>(mut_b, o2);
            }
            if (own) { mu_tend = amt_ldsv<T, VW>(D2 + 6 * TW + c); msfty_c = amt_ldsv<T, VW>(D2 + 1 * TW + c); }
            AMT_STAMP(2, 1);
            __syncthreads();                                     // 1: AB complete, D2/T1 row j no longer read
            AMT_STAMP(2, 2);
            V dmdt(T(0));
            if (own) {                                           // :147, sequential in k
                int k = 0;
                for (; k + AMT_CHAIN <= nk; k += AMT_CHAIN) {
                    V a[AMT_CHAIN];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) a[q] = s_dnw(k + q) * amt_ldsv<T, VW>(AB + (k + q) * TC + c);
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) dmdt = dmdt + a[q];
                }
                for (; k < nk; ++k) dmdt = dmdt + s_dnw(k) * amt_ldsv<T, VW>(AB + k * TC + c);
                // with dmdt: this row's mu_tend and msfty, for the cell waves' P2 and P3 (D2 moves on to
                // row j+1 below; these stay until the column wave passes barrier 1 of the next row)
                amt_stsv<T, VW>(DM + c, dmdt);
                amt_stsv<T, VW>(DM + TC + c, mu_tend);
                amt_stsv<T, VW>(DM + 2 * TC + c, msfty_c);
            }
            if (more) {                                          // install the 2-D row j+1
#pragma unroll
                for (int q = 0; q < N2D; ++q) {
                    if (inmem) amt_stsv<T, VW>(D2 + q * TW + c, d2v[q]);
                    if (halo_r) D2[q * TW + TC] = d2h[q];
                    if (tail) D2[q * TW + c] = d2h[q];
                }
            }
            AMT_STAMP(2, 3);
            amt_lds_barrier();                                   // 2: DM published
            AMT_STAMP(2, 4);
            AMT_STAMP(2, 5);
            amt_lds_barrier();                                   // 3: AB holds the increments
            AMT_STAMP(2, 6);
            if (own) {                                           // :161, sequential in k; AB[k] <- ww(k), the value BEFORE increment k
                V wwu = ww1in;
                int k = 0;
                for (; k + AMT_CHAIN <= nk; k += AMT_CHAIN) {
                    V b[AMT_CHAIN];
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) b[q] = amt_ldsv<T, VW>(AB + (k + q) * TC + c);
#pragma unroll
                    for (int q = 0; q < AMT_CHAIN; ++q) { amt_stsv<T, VW>(AB + (k + q) * TC + c, wwu); wwu = wwu - b[q]; }
                }
                for (; k < nk; ++k) { const V bk = amt_ldsv<T, VW>(AB + k * TC + c); amt_stsv<T, VW>(AB + k * TC + c, wwu); wwu = wwu - bk; }
            }
            AMT_STAMP(2, 7);
            __syncthreads();                                     // 4: ww of the recurrence published (DMA landed)
            // The 2-D mass update (:151-157) AFTER the second chain, although its inputs are ready since barrier 2: issued
            // there, its four stores were the newest vector-memory operations of the wave when the chain loop began, and
            // the loop's first LDS read into a register with a pending load waited vmcnt(0) -- for the stores' round trip
            // to memory, on the critical path of every cell wave parked at barrier 4.  Here their round trip ends
            // somewhere in the next row's P1.
            if (any) {
                const V mu_new = mu_old + dts * (dmdt + mu_tend);
                amt_stv<T, VW>(mu_b, o2, mu_new, all, on);
                amt_stv<T, VW>(mudf_b, o2, (dmdt + mu_tend), all, on);
                amt_stv<T, VW>(muts_b, o2, mut_v + mu_new, all, on);
                amt_stv<T, VW>(muave_b, o2, T(.5) * ((T(1.) + p.epssm) * mu_new + (T(1.) - p.epssm) * mu_old), all, on);
            }
        }
    } else {
        // =====================================================================
        // cell waves
        // =====================================================================
        const int il = lane % TI, h = lane / TI;
        const int c = il * VW;                             // first of my columns within the tile
        const int ii = col0 + c;
        const bool inmem = ii + VW <= p.idim;              // my columns lie inside the memory row
        bool on[VW], act = false, all = true;              // which of them are window columns
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            on[e] = (ii + e >= p.i0) && (ii + e <= p.i1);
            act = act || on[e]; all = all && on[e];
        }
        const int kfw = w * LW;                            // first level of this wave (zero-based)
        const int kf  = kfw + h * KPT;                     // first level of this lane
        // LDS indices are written (wave-uniform level row) * stride + (per-lane offset): the per-lane
        // part is ONE register for all [level][column] buffers (lc) and one for the per-level
        // records (lh), whatever the number of level groups
        const int lc = h * KPT * TC + c;                   // my level group's rows, my column
        const int lh = h * KPT;                            // my level group's first level within the wave
        const int nrw = FULL ? LW : (nk - kfw < LW ? nk - kfw : LW);     // real levels of this wave (>= 1)
        const bool has_above = (kf + KPT < nk);            // zero-based level kf+KPT exists
        const T rdx = p.rdx, rdy = p.rdy, dts = p.dts;
        const T hrdy = T(.5) * rdy, hrdx = T(.5) * rdx;

        // Wave-uniform base pointers (SGPR pairs), fixed for the whole march: element (column 0 of
        // the tile, the wave's first level, row ja) of every 3-D array.  The column, the level group,
        // the level step and the row advance go into an UNSIGNED 32-bit per-lane byte offset, which the
        // launcher keeps below 2^32 (amt_march_rows_for: max_rows).
        const long e3 = (long)ja * js + (long)(p.k1 + kfw) * idim + (long)col0;
        const T *u_b = p.u + e3, *u1_b = p.u_1 + e3, *ft_b = p.ft + e3, *ww1_b = p.ww_1 + e3;
        const T *t1_b = p.t_1 + e3, *v_b = p.v + e3, *v1_b = p.v_1 + e3;   // row ja
        T *t_b = p.t + e3, *tave_b = p.t_ave + e3, *ww_b = p.ww + e3;
        const unsigned vo = (unsigned)c * W;
        // byte offset of level slot m of this lane from the wave's first level
        const unsigned hoff = FULL ? (unsigned)(h * KPT) * lev : 0u;
        auto lo = [&](int m) -> unsigned {
            if (FULL) return (unsigned)m * lev;
            int L = h * KPT + m;
            L = L < nrw ? L : nrw - 1;
            return (unsigned)L * lev;
        };

        // ---------------- LDS-DMA machinery ----------------
        // Lane roles of the DMA instructions are recomputed from the lane id where they are used (`ln`:
        // the row loop passes a laundered copy, so that these row-invariant offsets are rebuilt with a
        // few VALU operations per row instead of occupying -- or spilling -- registers across P1).
        constexpr int EPL = 16 / (int)sizeof(T);           // elements per lane of one DMA instruction
        constexpr int LPL = TC / EPL;                      // lanes per level row
        constexpr int LPI = 64 / LPL;                      // levels per DMA instruction
        static_assert(!DMA || (LPL >= 1 && LPL <= 64), "a level row of the tile is at most one DMA instruction");
        // copies the wave's LW levels of j row (ja + rows) of `src` (uniform base at row ja) to lds
        auto dma_rows = [&](int ln, const T *src, int rows, T *lds) {
            const int dl = ln / LPL;                                         // level within the instruction
            const unsigned dch = (unsigned)(ln % LPL) * 16u;                 // my 16-byte chunk of the row
            const bool dok = col0 + (ln % LPL) * EPL < p.idim;          // chunk lies inside the memory row
            if constexpr (LW % LPI == 0) {
                // a wave fetches the levels it computes
#pragma unroll
                for (int q = 0; q < LW / LPI; ++q) {
                    const char *ub = reinterpret_cast<const char *>(src) + (FULL ? (size_t)(q * LPI) * lev : (size_t)0);
                    unsigned ro = (unsigned)dl * lev + dch + (unsigned)rows * row3;   // the row advance rides in the lane offset
                    if (!FULL) {                                                     // virtual levels read the last real one
                        const int l = q * LPI + dl;
                        ro = (unsigned)(l < nrw ? l : nrw - 1) * lev + dch + (unsigned)rows * row3;
                    }
                    if (dok)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + ro),
                                                         (__attribute__((address_space(3))) void *)(lds + (size_t)(kfw + q * LPI) * TC),
                                                         16, 0, AMT_NT_DMA);
                }
            } else {
                // levels per wave are not whole DMA instructions (3 or 5 levels per lane): the instructions of
                // the tile's nkr level rows are dealt round-robin to the cell waves, whatever levels they compute
                const int ninst = (nkr + LPI - 1) / LPI;
#pragma unroll
                for (int n = 0; n < (LW + LPI - 1) / LPI; ++n) {
                    const int q = w + n * nc;                                    // wave-uniform
                    if (q < ninst) {
                        const int g0 = q * LPI;                                  // first level row of the instruction
                        const int g0c = g0 < nk - 1 ? g0 : nk - 1;               // source level rows are clamped to the real ones
                        const char *ub = reinterpret_cast<const char *>(src) + ((long)g0c - (long)kfw) * (long)lev;
                        const int dmax = nk - 1 - g0c;
                        const unsigned ro = (unsigned)(dl < dmax ? dl : dmax) * lev + dch + (unsigned)rows * row3;
                        if (dok && g0 + dl < nkr)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + ro),
                                                             (__attribute__((address_space(3))) void *)(lds + (size_t)g0 * TC),
                                                             16, 0, AMT_NT_DMA);
                    }
                }
            }
        };
        // i halo of a t_1 row: per level the elements left of column 0 and right of column TC-1, one
        // dword per lane into TH[level][side]: 2*DPE lanes per level, LPH levels per instruction
        constexpr int DPE = (int)sizeof(T) / 4;            // dwords per element
        constexpr int LPH = 64 / (2 * DPE);
        auto dma_halo = [&](int ln, const T *src, int rows, T *lds) {
            const int hlv = ln / (2 * DPE), hside = (ln / DPE) & 1;
            const bool hmem = hside ? halo_r_mem : halo_l_mem;
            const unsigned hvo = (unsigned)(hside ? (TC + 1) * (int)sizeof(T) : 0) + (unsigned)(ln % DPE) * 4u;   // from element -1
#pragma unroll
            for (int q = 0; q < (LW + LPH - 1) / LPH; ++q) {
                const int l = q * LPH + hlv;
                const int lc = FULL ? l : (l < nrw ? l : nrw - 1);
                const unsigned ro = (unsigned)lc * lev + hvo + (unsigned)rows * row3;
                if (hmem && l < LW)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(src - 1) + ro),
                                                     (__attribute__((address_space(3))) void *)(lds + (size_t)(kfw + q * LPH) * 2),
                                                     4, 0, 0);
            }
        };
        // element TC (right of the tile) of a u / u_1 row: DPE lanes per level, one dword each, into UH[which][level]
        constexpr int LPU = 64 / DPE;
        auto dma_uhalo = [&](int ln, const T *src, int rows, T *lds) {
            const int ulv = ln / DPE;
            const unsigned uvo = (unsigned)TC * W + (unsigned)(ln % DPE) * 4u;
#pragma unroll
            for (int q = 0; q < (LW + LPU - 1) / LPU; ++q) {
                const int l = q * LPU + ulv;
                const int lc = FULL ? l : (l < nrw ? l : nrw - 1);
                const unsigned ro = (unsigned)lc * lev + uvo + (unsigned)rows * row3;
                if (halo_r_mem && l < LW)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(src) + ro),
                                                     (__attribute__((address_space(3))) void *)(lds + (size_t)(kfw + q * LPU)),
                                                     4, 0, 0);
            }
        };
        // the DMA set of one row advance: what P1 of row (ja + r) needs beyond what is already in LDS
        auto dma_next = [&](int ln, int r, T *t1dst, T *thdst) {
            dma_rows(ln, t1_b, r + 1, t1dst);                                    // t_1(j+1) of that row
            dma_rows(ln, v_b, r + 1, VB);                                        // v(j+1)
            dma_halo(ln, t1_b, r + 1, thdst);
            if (XD >= 1) dma_rows(ln, v1_b, r + 1, V1);                          // v_1(j+1)
            if (XD >= 2) dma_uhalo(ln, u_b, r, UH);                              // u, u_1 at column TC of row j
            if (XD >= 3) dma_uhalo(ln, u1_b, r, UH + nkr);
            if (XD >= 2) dma_rows(ln, u_b, r, U);                                // u(j)
            if (XD >= 3) dma_rows(ln, u1_b, r, U1);
        };
        // register flavour of the same: the tile-edge lanes fetch the i halo of a t_1 row
        const bool edge_l = (il == 0) && halo_l_mem, edge_r = (il == TI - 1) && halo_r_mem;
        auto reg_halo = [&](const T *src, unsigned off_row, T *thdst) {      // src: uniform base at (row, level kfw)
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                const unsigned om = off_row + hoff + lo(m);
                if (edge_l) thdst[(kfw + m) * 2 + 2 * lh] = amt_ld(src - 1, om);
                if (edge_r) thdst[(kfw + m) * 2 + 2 * lh + 1] = amt_ld(src + TC, om);
            }
        };

        // carried in registers from row to row (per owned level): the two j-face fluxes
        V vfm[KPT], vft[KPT];
#pragma unroll
        for (int m = 0; m < KPT; ++m) { vfm[m] = V(T(0)); vft[m] = V(T(0)); }

        // P1's global loads of a batch of levels (amt_p1_batch).  (Issuing the whole lane's batch one row ahead, right behind
        // barrier 4 and ahead of P3's stores, was built as well: 1-2 % SLOWER on every shape -- profiles/r04_80level.md.)
        constexpr int PB = amt_p1_batch<T, VW, KPT, HL, XD, DMA, WM>();
        V g_v1[KPT], g_uu[KPT], g_u1[KPT];
        T g_un[KPT], g_u1n[KPT];
        auto p1_loads = [&](int m0, int m1, unsigned orow) {     // orow: per-lane byte offset of the row (lane part included)
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                if (q < m0 || q >= m1) continue;
                const unsigned oq = orow + lo(q);
                if (XD < 1) g_v1[q] = amt_ldv<T, VW>(v1_b + js, oq);
                if (XD < 2) { g_uu[q] = amt_ldv_stream<2, NTL, T, VW>(u_b, oq); g_un[q] = amt_ld_stream<2, NTL>(u_b + VW, oq); }
                if (XD < 3) { g_u1[q] = amt_ldv_stream<2, NTL, T, VW>(u1_b, oq); g_u1n[q] = amt_ld_stream<2, NTL>(u1_b + VW, oq); }
            }
        };

        // ---- prologue: everything P1 of row ja needs; j-face fluxes of row ja ----
        if (DMA) {
            dma_rows(lane, t1_b, 0, T1);
            dma_halo(lane, t1_b, 0, TH);
            dma_next(lane, 0, T1 + t1buf, TH + thbuf);
        } else {
            reg_halo(t1_b, 0u, TH);
        }
        {
            V muv_j(T(0)), mvx_j(T(0));
            if (act) { muv_j = amt_ldv<T, VW>(p.muv + e2, vo); mvx_j = amt_ldv<T, VW>(p.msfvx_inv + e2, vo); }
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                const unsigned om = vo + hoff + lo(m);
                if (!DMA && inmem) amt_stsv<T, VW>(T1 + (kfw + m) * TC + lc, amt_ldv<T, VW>(t1_b, om));
                if (act) {
                    const V vv = amt_ldv<T, VW>(v_b, om);
                    vfm[m] = vv + muv_j * amt_ldv<T, VW>(v1_b, om) * mvx_j;
                    vft[m] = vv * (amt_ldv<T, VW>(t1_b, om) + amt_ldv<T, VW>(t1_b - js, om));
                }
            }
        }
        __syncthreads();                               // everything staged, DMA landed

        unsigned o3 = vo + hoff;                       // per-lane byte offset of the current row
        for (int jj = ja; jj <= jb; ++jj, o3 += row3) {
            V hf[KPT], tw[KPT];
            const int par = (jj - ja) & 1;
            const T *T1c = T1 + par * t1buf;                         // t_1 row j
            T *T1n = T1 + (par ^ 1) * t1buf;                         // t_1 row j+1 (DMA'd during the previous row / written now)
            const T *THc = TH + par * thbuf;
            T *THn = TH + (par ^ 1) * thbuf;
            const bool more = (jj < jb);
            AMT_STAMP(stamp_slot, 0);

            // ---------------- P1: per-cell work from pure inputs ----------------
            if (!DMA) {
                reg_halo(t1_b + js, o3 - vo - hoff, THn);                      // i halo of t_1 row j+1
                if (inmem && !act) {                                 // columns of the tile outside the window
#pragma unroll
                    for (int m = 0; m < KPT; ++m)
                        amt_stsv<T, VW>(T1n + (kfw + m) * TC + lc, amt_ldv<T, VW>(t1_b + js, o3 + lo(m)));
                }
            }
            if (act) {
                const V msftx = amt_ldsv<T, VW>(D2 + 0 * TW + c);
                const V msfty_p1 = amt_ldsv<T, VW>(D2 + 1 * TW + c);
                const V mm = msftx * msfty_p1;
                const V muu_i = amt_ldsv<T, VW>(D2 + 2 * TW + c), muu_ip = amt_ldsv<T, VW>(D2 + 2 * TW + c + 1);
                const V msfuy_i = amt_ldsv<T, VW>(D2 + 3 * TW + c), msfuy_ip = amt_ldsv<T, VW>(D2 + 3 * TW + c + 1);
                const V muv_p = amt_ldsv<T, VW>(D2 + 4 * TW + c), mvx_p = amt_ldsv<T, VW>(D2 + 5 * TW + c);
                // Neighbour columns c-1 and c+VW are read unclamped: for the tile's first / last lane they
                // fall into the adjacent LDS row (always inside the LDS image) and the halo value is
                // selected instead -- one address register serves all three reads.
                // The global loads of P1 (what does not ride the LDS-DMA: v_1(j+1), u and u_1 at i and i+1), PB levels at a
                // time: all loads of a batch are issued before any is consumed (the compiler barrier keeps them above the
                // batch's LDS stores), so a batch pays ONE memory latency.  Left to itself hipcc interleaves load, wait and
                // use level by level -- lowest register pressure, one full latency per level: four per row in the 60-level
                // fp64 shape (ISA of r03: s_waitcnt vmcnt(0) after each level's three loads).
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    if (PB > 1 && m % PB == 0) {
                        p1_loads(m, m + PB, o3);
                        asm volatile("" ::: "memory");
                    }
                    const unsigned om = o3 + lo(m);
                    const int K = kf + m;
                    V vn, v1n, t1n, uu, u1;
                    T un, u1n;                                               // u, u_1 at my last column + 1
                    if (DMA) { vn = amt_ldsv<T, VW>(VB + (kfw + m) * TC + lc); t1n = amt_ldsv<T, VW>(T1n + (kfw + m) * TC + lc); }
                    else {
                        vn = amt_ldv<T, VW>(v_b + js, om); t1n = amt_ldv<T, VW>(t1_b + js, om);
                        amt_stsv<T, VW>(T1n + (kfw + m) * TC + lc, t1n);
                    }
                    if (XD >= 1) v1n = amt_ldsv<T, VW>(V1 + (kfw + m) * TC + lc); else if (PB > 1) v1n = g_v1[m]; else v1n = amt_ldv<T, VW>(v1_b + js, om);
                    if (XD >= 2) {
                        uu = amt_ldsv<T, VW>(U + (kfw + m) * TC + lc);
                        const T in = U[(kfw + m) * TC + lc + VW];
                        un = (il == TI - 1) ? UH[kfw + m + lh] : in;
                    } else if (PB > 1) { uu = g_uu[m]; un = g_un[m]; }
                    else { uu = amt_ldv_stream<2, NTL, T, VW>(u_b, om); un = amt_ld_stream<2, NTL>(u_b + VW, om); }
                    if (XD >= 3) {
                        u1 = amt_ldsv<T, VW>(U1 + (kfw + m) * TC + lc);
                        const T in = U1[(kfw + m) * TC + lc + VW];
                        u1n = (il == TI - 1) ? UH[nkr + kfw + m + lh] : in;
                    } else if (PB > 1) { u1 = g_u1[m]; u1n = g_u1n[m]; }
                    else { u1 = amt_ldv_stream<2, NTL, T, VW>(u1_b, om); u1n = amt_ld_stream<2, NTL>(u1_b + VW, om); }
                    const V t1c = amt_ldsv<T, VW>(T1c + (kfw + m) * TC + lc);
                    const T tl_in = (T1c + (kfw + m) * TC + lc)[-1], tr_in = T1c[(kfw + m) * TC + lc + VW];
                    const T tl = (il == 0) ? THc[(kfw + m) * 2 + 2 * lh] : tl_in;
                    const T tr = (il == TI - 1) ? THc[(kfw + m) * 2 + 2 * lh + 1] : tr_in;
                    V uup, u1p, t1l, t1r;                                    // the same fields one column to the right / left
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        uup.x[e] = e + 1 < VW ? uu.x[e + 1 < VW ? e + 1 : e] : un;
                        u1p.x[e] = e + 1 < VW ? u1.x[e + 1 < VW ? e + 1 : e] : u1n;
                        t1r.x[e] = e + 1 < VW ? t1c.x[e + 1 < VW ? e + 1 : e] : tr;
                        t1l.x[e] = e > 0 ? t1c.x[e > 0 ? e - 1 : 0] : tl;
                    }
                    // :142-146
                    const V vfm_n = vn + muv_p * v1n * mvx_p;
                    const V d = mm * ( rdy * (vfm_n - vfm[m])
                                     + rdx * ( (uup + muu_ip * u1p / msfuy_ip)
                                             - (uu  + muu_i  * u1  / msfuy_i ) ));
                    amt_stsv<T, VW>(AB + (kfw + m) * TC + lc, d);              // dnw(k)*d, the term of :147, is formed by the column wave
                    // horizontal part of :237-245
                    const V vft_n = vn * (t1n + t1c);
                    hf[m] = msftx * ( hrdy * (vft_n - vft[m])
                                    + hrdx * ( uup * (t1r + t1c) - uu * (t1c + t1l) ) );
                    // fnm(k)*t_1(k) + fnp(k)*t_1(k-1) of :227 (unused for Fortran level 1)
                    const V t1km1 = (K > 0) ? amt_ldsv<T, VW>(T1c + (kfw + m) * TC + lc - (K > 0 ? TC : 0)) : V(T(0));
                    tw[m] = S1[4 * (kfw + m) + 4 * lh + 1] * t1c + S1[4 * (kfw + m) + 4 * lh + 2] * t1km1;
                    vfm[m] = vfm_n; vft[m] = vft_n;                   // the faces of row j+1
                }
                // wdtn at the level above a lane's last one (:227) needs fnm*t_1(k)+fnp*t_1(k-1) of that
                // level, which is tw[0] of the level group above: published here, read in P3 (two row
                // parities: the group above may be a row ahead by then)
                amt_stsv<T, VW>(TWB + (par * nc * HL + w * HL) * TC + lane * VW, tw[0]);   // [group w*HL+h][column c] = lane*VW
            }
            AMT_STAMP(stamp_slot, 1);
            __syncthreads();                                         // 1: AB complete; row j of T1/TH/VB/D2 dead
            AMT_STAMP(stamp_slot, 2);

            if (DMA && more) {   // what the next row's P1 needs: no registers, lands before barrier 4
                int ln = lane;
                asm volatile("" : "+v"(ln));
                dma_next(ln, jj - ja + 1, T1 + par * t1buf, TH + par * thbuf);
            }
            // while the column wave sums dmdt: issue the loads that only P3 consumes
            V told[KPT], ftk[KPT], w1[KPT];
            V w1_above(T(0));
            if (act) {
                // the level above my last one is a real level (has_above): never clamped
                if (has_above) w1_above = amt_ldv<T, VW>(ww1_b, o3 + (unsigned)((FULL ? 0 : h * KPT) + KPT) * lev);
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + lo(m);
                    told[m] = amt_ldv_stream<1, NTL, T, VW>(t_b, om);
                    ftk[m] = amt_ldv_stream<1, NTL, T, VW>(ft_b, om);
                    w1[m] = amt_ldv_stream<1, NTL, T, VW>(ww1_b, om);
                }
            }
            AMT_STAMP(stamp_slot, 3);
            amt_lds_barrier();                                       // 2: DM published (DMA keeps flying)
            AMT_STAMP(stamp_slot, 4);
            V inc_last(T(0));                                       // my top level's increment (:161)
            if (act) {
                const V dmdt = amt_ldsv<T, VW>(DM + c);
                const V mu_tend = amt_ldsv<T, VW>(DM + TC + c);
                const V msfty = amt_ldsv<T, VW>(DM + 2 * TC + c);
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const V inc = S1[4 * (kfw + m) + 4 * lh] * (dmdt + amt_ldsv<T, VW>(AB + (kfw + m) * TC + lc) + mu_tend) / msfty;   // :161
                    amt_stsv<T, VW>(AB + (kfw + m) * TC + lc, inc);
                    if (m == KPT - 1) inc_last = inc;
                }
            }
            AMT_STAMP(stamp_slot, 5);
            amt_lds_barrier();                                       // 3: AB holds the increments
            if (AMT_TAVE_EARLY && act) {
                // :211  t_ave = t as it came in.  Of P3's three stores per level this one waits for no chain, only for its own
                // load (issued behind barrier 1): it goes out here, under the second chain, and P3 keeps two stores per level
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const bool real = FULL || kf + m < nk;
                    bool onr[VW];
#pragma unroll
                    for (int e = 0; e < VW; ++e) onr[e] = on[e] && real;
                    amt_stv<T, VW>(tave_b, o3 + lo(m), told[m], all && real, onr);
                }
            }
            __syncthreads();                                         // 4: ww of the recurrence published; DMA landed
            AMT_STAMP(stamp_slot, 6);

            // ---------------- P3: vertical flux, theta ----------------
            if (act) {
                const V msfty = amt_ldsv<T, VW>(DM + 2 * TC + c);
                V wwu = amt_ldsv<T, VW>(AB + kfw * TC + lc);           // ww of :161 at my first level
                V wd_k = (kf == 0) ? V(T(0)) : (wwu - w1[0]) * tw[0];   // wdtn(i,1) = 0 (:220)
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    const unsigned om = o3 + lo(m);
                    const int K = kf + m;
                    const bool real = FULL || K < nk;
                    bool onr[VW];
#pragma unroll
                    for (int e = 0; e < VW; ++e) onr[e] = on[e] && real;
                    const bool allr = all && real;
                    const V wout = wwu - w1[m];                      // :170
                    amt_stv<T, VW>(ww_b, om, wout, allr, onr);
                    // wdtn at level K+1 (:221,:227)
                    V wd_n(T(0));
                    const V wwu_n = (m + 1 < KPT) ? amt_ldsv<T, VW>(AB + (kfw + (m + 1 < KPT ? m + 1 : m)) * TC + lc) : wwu - inc_last;
                    if (m + 1 < KPT) wd_n = (wwu_n - w1[m + 1 < KPT ? m + 1 : 0]) * tw[m + 1 < KPT ? m + 1 : 0];
                    else if (has_above) wd_n = (wwu_n - w1_above) * amt_ldsv<T, VW>(TWB + (par * nc * HL + w * HL + 1) * TC + lane * VW);
                    if (!FULL && K + 1 >= nk) wd_n = V(T(0));        // wdtn(kde) = 0, :221
                    if (!AMT_TAVE_EARLY) amt_stv<T, VW>(tave_b, om, told[m], allr, onr);  // :211 (stored under the second chain otherwise)
                    const V tb = told[m] + msfty * dts * ftk[m];     // :212
                    amt_stv<T, VW>(t_b, om, tb - dts * msfty * ( hf[m] + S1[4 * (kfw + m) + 4 * lh + 3] * (wd_n - wd_k) ), allr, onr);   // :237-246
                    wwu = wwu_n; wd_k = wd_n;
                }
            }
            AMT_STAMP(stamp_slot, 7);
            // No barrier here.  What the next row's P1 writes (its own AB slots, and -- register
            // flavour -- the T1/TH buffer that was READ in this row's P1) is read by no other wave
            // before barrier 1 of the next row.
        }
        AMT_SPAN(1);
    }
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
struct AmtMarchShape {
    int vw, kpt, hl, xd;
    bool dma;
    int wm;          // most waves per workgroup (16 or 12)
};

static size_t amt_march_lds_bytes(int wbytes, const AmtMarchShape &s, int nk)
{
    const int lw = s.kpt * s.hl, tc = (64 / s.hl) * s.vw;
    const size_t nkr = (size_t)((nk + lw - 1) / lw) * lw;          // nk rounded up to whole cell waves
    // AB, T1[2] (+ VB) (+ V1, U, U1): [nkr][tc]; TH [2][nkr][2] (+ UH [2][nkr]); D2 [7][tc+2]; DM [3][tc]; S1 [nkr][4]; TWB [2][nkr/kpt][tc]
    const size_t bufs = 3 + (s.dma ? 1 : 0) + s.xd;
    return (bufs * nkr * tc + 4 * nkr + (s.dma ? 2 * nkr : 0) + (size_t)AMT_N2D * (tc + 2) + 3 * tc + 4 * nkr + 2 * (nkr / s.kpt) * tc) * (size_t)wbytes;
}

static int amt_march_waves(const AmtMarchShape &s, int nk)
{
    const int lw = s.kpt * s.hl;
    return (nk + lw - 1) / lw + 1;                                 // cell waves + the column wave
}

// The LDS-DMA flavour moves 16-byte chunks per lane, but `global_load_lds_dwordx4` needs no alignment
// of its global SOURCE (measured on gfx950: sources shifted by 4, 8 or 12 bytes copy correctly and at
// the aligned rate, 6.1-6.2 TB/s; profiles/r02_lds_dma_alignment.md), so any WRF layout -- odd row
// lengths, arrays that start anywhere -- takes it.  A chunk that starts inside the memory row but runs
// past its end (tiles anchored at an odd window column, or a row length that is no multiple of a chunk)
// reads on into the next level row of the same array and fills LDS columns no lane uses: the last level
// the DMA fetches is the window's last one, so the array must hold one more level row behind it
// (kme > k_end, which the C-ABI's precondition kme >= kte = k_end + 1 gives every caller; checked here
// because the kernel's safety rests on it), otherwise the register flavour runs.
template <typename T> static bool amt_march_dma_layout_ok(const AmtParams<T> &p)
{
    return p.k1 + p.nk < p.kdim;
}

static bool amt_march_shape_valid(int wbytes, const AmtMarchShape &s)
{
    if (s.vw != 1 && s.vw != 2) return false;
    if (s.hl != 1 && s.hl != 2 && s.hl != 4) return false;
    if (s.kpt < 2 || s.kpt > 8 || s.kpt == 7) return false;
    if (s.wm != 16 && s.wm != 12) return false;
    if (s.xd < 0 || s.xd > 3 || (s.xd && !s.dma)) return false;
    if (s.dma) {
        const int tc = (64 / s.hl) * s.vw, epl = 16 / wbytes, lpl = tc / epl;
        if (lpl < 1 || lpl > 64) return false;                         // a level row is at most one DMA instruction
    }
    return true;
}

// The FULL = true build of a shape skips the clamping of virtual levels; where that build needs
// scratch and the general one does not (fp64, 4 levels per lane with level groups at 16 waves: 12 B/lane against 0),
// the general build also runs the level counts that fill the waves.
static bool amt_march_full_build_ok(int wbytes, const AmtMarchShape &s)
{
    return !(wbytes == 8 && s.hl > 1 && s.wm == 16 && s.kpt >= 4);
}

static bool amt_march_shape_feasible(int wbytes, const AmtMarchShape &s, int nk)
{
    if (!amt_march_shape_valid(wbytes, s)) return false;
    if (amt_march_waves(s, nk) > s.wm) return false;
    return amt_march_lds_bytes(wbytes, s, nk) <= 160 * 1024;
}

// Every instantiation the library carries.  X(T, VW, KPT, HL, XD, DMA, WM)   (-DAMT_MARCH_SHAPES=... or -include: a short list for
// ISA inspection builds)
#ifndef AMT_MARCH_SHAPES
#define AMT_MARCH_SHAPES(X)                                                                      \
    /* fp64: 4 levels per lane at 16 waves, up to 6 at 12 waves; 2 for nk <= 30 */               \
    X(double, 1, 2, 1, 0, true, 16) X(double, 1, 2, 1, 3, true, 16) X(double, 1, 2, 1, 0, false, 16)    \
    X(double, 1, 4, 1, 0, true, 16) X(double, 1, 4, 1, 1, true, 16) X(double, 1, 4, 1, 0, false, 16)    \
    X(double, 1, 4, 2, 0, true, 16) X(double, 1, 4, 2, 0, false, 16)                             \
    X(double, 1, 4, 2, 0, true, 12) X(double, 1, 4, 2, 1, true, 12) X(double, 1, 4, 2, 0, false, 12)    \
    X(double, 1, 6, 2, 0, true, 12) X(double, 1, 6, 2, 0, false, 12)                             \
    X(double, 1, 3, 1, 0, true, 16) X(double, 1, 3, 2, 0, true, 16)                              \
    X(double, 1, 2, 2, 0, true, 16) X(double, 1, 2, 2, 3, true, 16)                              \
    X(double, 1, 4, 4, 0, true, 16) X(double, 1, 4, 4, 0, false, 16)                             \
    X(double, 1, 4, 4, 0, true, 12) X(double, 1, 4, 4, 0, false, 12)                             \
    X(double, 1, 6, 4, 0, true, 12) X(double, 1, 6, 4, 0, false, 12)                             \
    /* fp32, one column per lane (4-byte accesses; any row length) */                            \
    X(float, 1, 4, 1, 0, true, 16) X(float, 1, 4, 1, 3, true, 16) X(float, 1, 4, 1, 0, false, 16)       \
    X(float, 1, 8, 1, 0, true, 12) X(float, 1, 8, 1, 0, false, 12)                               \
    X(float, 1, 4, 2, 0, true, 16) X(float, 1, 4, 2, 0, false, 16)                               \
    X(float, 1, 8, 2, 0, true, 12) X(float, 1, 8, 2, 0, false, 12)                               \
    X(float, 1, 8, 4, 0, true, 12) X(float, 1, 8, 4, 0, false, 12)                               \
    /* fp32, two columns per lane (8-byte accesses: the fp64 profile; even row length) */        \
    X(float, 2, 2, 1, 0, true, 16) X(float, 2, 2, 1, 3, true, 16)                                \
    X(float, 2, 4, 1, 0, true, 16) X(float, 2, 4, 1, 1, true, 16) X(float, 2, 4, 1, 0, false, 16)       \
    X(float, 2, 4, 2, 0, true, 16) X(float, 2, 4, 2, 0, false, 16)                               \
    X(float, 2, 4, 2, 0, true, 12) X(float, 2, 4, 2, 1, true, 12) X(float, 2, 4, 2, 0, false, 12)       \
    X(float, 2, 6, 2, 0, true, 12) X(float, 2, 6, 2, 0, false, 12)                               \
    X(float, 2, 3, 1, 0, true, 16) X(float, 2, 3, 2, 0, true, 16)                                \
    X(float, 2, 4, 4, 0, true, 16) X(float, 2, 4, 4, 0, false, 16)                               \
    X(float, 2, 4, 4, 0, true, 12) X(float, 2, 4, 4, 0, false, 12)                               \
    X(float, 2, 6, 4, 0, true, 12) X(float, 2, 6, 4, 0, false, 12)
#endif

template <typename T> struct AmtMarchEntry {
    AmtMarchShape shape;
    const void *kernel[4];                                         // [2 * cached + FULL]
    unsigned lds_granted[4];                                       // bit d: device d was told this kernel may use all of LDS
    void (*launch[4])(hipStream_t, const AmtParams<T> &, const AmtMarchGrid &, int nw, size_t lds);
    const char *name, *name_cached;
};

template <typename T, int VW, int KPT, int HL, int XD, bool FULL, bool DMA, int WM, int NTL>
static void amt_march_launch_one(hipStream_t stream, const AmtParams<T> &p, const AmtMarchGrid &g, int nw, size_t lds)
{
    hipLaunchKernelGGL((amt_march_kernel<T, VW, KPT, HL, XD, FULL, DMA, WM, NTL>), dim3(g.nwg), dim3(nw * 64), lds, stream, p, g);
}

template <typename T> static AmtMarchEntry<T> *amt_march_table(int *n);

#define AMT_ENTRY_IF(TT, VW, KPT, HL, XD, DMA, WM)                                                          \
    if constexpr (std::is_same<T, TT>::value)                                                               \
        tab[cnt++] = AmtMarchEntry<T>{AmtMarchShape{VW, KPT, HL, XD, DMA, WM},                                \
            {reinterpret_cast<const void *>(amt_march_kernel<TT, VW, KPT, HL, XD, false, DMA, WM, AMT_NT_LOAD>),     \
             reinterpret_cast<const void *>(amt_march_kernel<TT, VW, KPT, HL, XD, true, DMA, WM, AMT_NT_LOAD>),      \
             reinterpret_cast<const void *>(amt_march_kernel<TT, VW, KPT, HL, XD, false, DMA, WM, 0>),               \
             reinterpret_cast<const void *>(amt_march_kernel<TT, VW, KPT, HL, XD, true, DMA, WM, 0>)}, {0u, 0u, 0u, 0u}, \
            {amt_march_launch_one<TT, VW, KPT, HL, XD, false, DMA, WM, AMT_NT_LOAD>, amt_march_launch_one<TT, VW, KPT, HL, XD, true, DMA, WM, AMT_NT_LOAD>, \
             amt_march_launch_one<TT, VW, KPT, HL, XD, false, DMA, WM, 0>, amt_march_launch_one<TT, VW, KPT, HL, XD, true, DMA, WM, 0>}, \
            "amt_march_kernel<" #TT ", " #VW ", " #KPT ", " #HL ", " #XD ", FULL, " #DMA ", " #WM ", nt>",    \
            "amt_march_kernel<" #TT ", " #VW ", " #KPT ", " #HL ", " #XD ", FULL, " #DMA ", " #WM ", cached>"};

template <typename T> static AmtMarchEntry<T> *amt_march_table(int *n)
{
    static AmtMarchEntry<T> tab[64];
    static int cnt = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        AMT_MARCH_SHAPES(AMT_ENTRY_IF)
    });
    *n = cnt;
    return tab;
}

template <typename T> static AmtMarchEntry<T> *amt_march_find(const AmtMarchShape &s)
{
    int n = 0;
    AmtMarchEntry<T> *tab = amt_march_table<T>(&n);
    for (int i = 0; i < n; ++i) {
        const AmtMarchShape &t = tab[i].shape;
        if (t.vw == s.vw && t.kpt == s.kpt && t.hl == s.hl && t.xd == s.xd && t.dma == s.dma && t.wm == s.wm) return &tab[i];
    }
    return nullptr;
}

// Shape overrides: environment knobs, read once per process, or set at run
// time through amt_march_force_shape -- tuning, A/B timing in one process and the parity tests
// that walk every instantiation; 0 / -1 leave a parameter to the launcher.
struct AmtMarchEnv {
    int dma, kpt, hl, vw, xd, jrows, verbose, wm, xchunk;
    int beside_rounds;     // least rounds of workgroups of a launch beside another stream's kernels (AmtParams::edges == 2)
    int beside_reserve;    // compute units such a launch is planned to leave free in every round
    int nt;                // once-read streams: -1 by the row length (non-temporal where rows are whole 128-byte lines), 0 cached, 1 non-temporal
};
static AmtMarchEnv g_march_env = {1, 0, 0, 0, -1, 0, 0, 0, 0, 2, 0, -1};
// The instantiation the last plan of the calling thread chose (diagnosis / tests): "" before any launch.
static thread_local char g_march_last[360] = "";
extern "C" const char *amt_march_last_kernel(void) { return g_march_last; }
void amt_march_note_kernel(const char *name) { snprintf(g_march_last, sizeof g_march_last, "%s", name); }   // amt_api.hip: column kernel
static std::atomic<int> g_march_generation{0};     // bumped by amt_march_force_shape: cached plans are stale (host threads plan concurrently)
static std::once_flag g_march_env_once;
static const AmtMarchEnv &amt_march_env()
{
    std::call_once(g_march_env_once, [] {
        g_march_env = {amt_env_int("AMT_MARCH_DMA", 1), amt_env_int("AMT_MARCH_KPT", 0),
                       amt_env_int("AMT_MARCH_HL", 0), amt_env_int("AMT_MARCH_VW", 0),
                       amt_env_int("AMT_MARCH_XD", -1), amt_env_int("AMT_MARCH_JROWS", 0),
                       amt_env_int("AMT_MARCH_VERBOSE", 0), amt_env_int("AMT_MARCH_WM", 0),
                       amt_env_int("AMT_MARCH_XCHUNK", 0), amt_env_int("AMT_MARCH_BESIDE_ROUNDS", 2),
                       amt_env_int("AMT_MARCH_BESIDE_RESERVE", 0), amt_env_int("AMT_MARCH_NT", -1)};
        if (g_march_env.beside_rounds < 1) g_march_env.beside_rounds = 1;
        if (g_march_env.beside_reserve < 0) g_march_env.beside_reserve = 0;
    });
    return g_march_env;
}

extern "C" int amt_march_force_shape(int vw, int kpt, int hl, int xd, int dma, int jrows, int wm)
{
    (void)amt_march_env();
    g_march_env.vw = vw > 0 ? vw : 0;
    g_march_env.kpt = kpt > 0 ? kpt : 0;
    g_march_env.hl = hl > 0 ? hl : 0;
    g_march_env.xd = xd >= 0 ? xd : -1;
    g_march_env.dma = dma != 0;
    g_march_env.jrows = jrows > 0 ? jrows : 0;
    g_march_env.wm = wm > 0 ? wm : 0;
    ++g_march_generation;
    return 0;
}

extern "C" int amt_march_set_stream_policy(int policy)
{
    (void)amt_march_env();
    g_march_env.nt = policy < 0 ? -1 : policy ? 1 : 0;
    ++g_march_generation;
    return 0;
}

extern "C" int amt_march_set_beside(int rounds, int reserve_cus)
{
    (void)amt_march_env();
    g_march_env.beside_rounds = rounds > 0 ? rounds : 2;
    g_march_env.beside_reserve = reserve_cus > 0 ? reserve_cus : 0;
    ++g_march_generation;
    return 0;
}

// compute units a launch is planned for: all of them, or -- beside another stream's kernels -- all but the reserve
static int amt_march_plan_cus(int cus, int edges)
{
    if (edges != 2) return cus;
    const int left = cus - amt_march_env().beside_reserve;
    return left > cus / 2 ? left : cus / 2 > 0 ? cus / 2 : 1;
}

extern "C" int amt_march_set_xchunk(int xchunk)
{
    (void)amt_march_env();
    g_march_env.xchunk = xchunk > 0 ? xchunk : 0;
    ++g_march_generation;
    return 0;
}

template <typename T> static long amt_march_max_rows(const AmtParams<T> &p)
{
    // The per-lane byte offsets of the march are 32-bit and UNSIGNED: every access is (uniform 64-bit base) + (unsigned
    // offset, zero-extended) -- `global_load v, v_off, s[base:base+1]` -- and the offsets only ever grow from the block's
    // first row, so a block may span 4 GiB less the rows and levels read ahead (tests/test_gpu_13_fullsize.py runs blocks
    // whose offsets pass 2 GiB against the oracle).
    const long row_bytes = p.jstride * (long)sizeof(T);
    if (row_bytes <= 0) return 1L << 30;             // no rows to step over (a shape query without a domain)
    return ((1L << 32) - 40L * p.idim * (long)sizeof(T)) / row_bytes - 3;
}

// Rows per workgroup.  A block costs its rows plus a prologue (4 extra array-rows of loads, about
// half a row of time); workgroups run in rounds of one per CU (two small workgroups that share a CU
// each run slower than one with twice the rows: 128x60x128 fp64 28 us against 23), so the sweep takes
// about  rounds(r) * (r + 0.5)  row-times.  The r that minimises it: for a 510-row j-slab (8 GPUs)
// r = 32 would leave the last round 6 % full (1040 workgroups on 256 CUs) and cost 17 % more than
// r = 11.  Small launches get short blocks, down to one row (more workgroups: 64x40x64 takes 12 us
// with r = 1, 27 us with r = 4).
// How long a block may be (profiles/r04_rows.md): r03 capped it at 64 rows.  Recorded per workgroup (profiles/spans.py), a
// 16-round launch pays 16 prologues per CU and ends 110-200 us after its median CU (a block is about 1 % slower on some
// XCDs than on others, and every XCD gets one eighth of the blocks); dispatch gaps are 1 us.  Fewer, longer blocks save the
// prologues the model counts (4096x60x8192 fp64 512 rows -0.3 %, 8192x80x8192 fp32 683 rows = 6 whole rounds -0.7 %
// against 128 rows), and a launch that is ONE round -- as many blocks as CUs, each as long as it takes -- saves five times
// that: 4096x60x4096 fp64 1024 rows -1.3 .. -1.6 %, 4096x60x512 128 rows -2.0 .. -3.0 %, 8192x80x2048 fp32 -2.8 .. -3.9 %.
// So: no cap beyond what the 32-bit row offsets span (`max_rows`) -- except for the fp64 shapes with level groups
// (32-column tiles), which stay at 64 rows: at 128 they take the same time and read 1.8 % more (neighbouring tiles drift
// apart over a long block and lose each other's halo lines in L2: 46.19 against 45.37 GB for 4096x80x2048), at 256 and
// more they are 1 .. 2 % slower, in rounds or as one round.
static int amt_march_rows_cap(int wbytes, int hl) { return (wbytes == 8 && hl >= 2) ? 64 : 1 << 30; }
static int amt_march_rows(long ntile_i, int nj, int cus, long max_rows, int cap, int min_rounds, double *cost_out, long *rounds_out)
{
    double best = 1e300;
    int jrows = 1;
    long brounds = 1;
    for (int r = 1; r <= nj && r <= max_rows && r <= cap; ++r) {
        const long blocks = ntile_i * ((nj + r - 1) / r);
        const long rounds = (blocks + cus - 1) / cus;
        if (rounds < min_rounds && r > 1) continue;      // a launch beside another stream's kernels (AmtParams::edges == 2)
        const double cost = (double)rounds * (r + 0.5);
        if (cost < best - 1e-9 || (cost < best + 1e-9 && r > jrows)) { best = cost; jrows = r; brounds = rounds; }
    }
    if (cost_out) *cost_out = best;
    if (rounds_out) *rounds_out = brounds;
    return jrows;
}

// the row-count rule alone, for the host-logic tests (tests/test_march_rows.py)
extern "C" int amt_march_rows_for(long ntile_i, int nj, int cus, long max_rows, int wbytes, int hl)
{
    if (ntile_i < 1 || nj < 1 || cus < 1 || max_rows < 1) return 0;      // nothing to plan
    return amt_march_rows(ntile_i, nj, cus, max_rows, amt_march_rows_cap(wbytes, hl), 1, nullptr, nullptr);
}

// Shape preference.  Measured (profiles/r02_shapes.md): most waves to hide latency and fewest
// registers per lane first -- 4 levels per lane is what fits 127 VGPRs in fp64 (and in fp32 with two
// columns per lane) without scratch; more levels come from splitting the wave into level groups
// (HL), never from more levels per lane.  The DMA flavour wherever the layout allows it.
// Column 0 of the first i tile: the window's first column rounded down to a 128-byte line of the memory
// row (16 fp64 / 32 fp32 elements), so that a tile's plain loads and stores start on a line whenever the
// rows themselves do.
static int amt_march_cus(int dev);
// Rows that are not whole lines (WRF's own ims:ime = its-1:ite+1: 4098 elements) put every level row at
// another phase of the line anyway: there the tiles start AT the window, which saves the 65th tile that a
// window starting one element into the row would otherwise leave with a single column.
template <typename T> static int amt_march_col_lo(const AmtParams<T> &p)
{
    const int line = 128 / (int)sizeof(T);
    if (p.idim % line != 0) return p.i0;
    return p.i0 / line * line;
}

template <typename T> static bool amt_march_pick(const AmtParams<T> &p, AmtMarchShape &out)
{
    const int nk = p.nk;
    if (nk < 1) return false;
    const AmtMarchEnv &env = amt_march_env();
    const bool dma_ok = env.dma != 0 && amt_march_dma_layout_ok(p);
    const int wb = (int)sizeof(T);
    // only instantiations without scratch are listed (tests/test_kernel_resources.py checks the build).
    // Order: as many cell waves as the 16-wave budget gives (memory-level parallelism; profiles/r02_shapes.md:
    // 3 levels per lane beat 4 by 1-2.4 % wherever they add waves), then the 12-wave builds.
    static const AmtMarchShape pref64[] = {
        {1, 2, 1, 3, true, 16}, {1, 3, 1, 0, true, 16}, {1, 4, 1, 0, true, 16}, {1, 3, 2, 0, true, 16}, {1, 4, 2, 0, true, 12},
        {1, 4, 2, 0, true, 16}, {1, 6, 2, 0, true, 12},
        {1, 4, 4, 0, true, 12}, {1, 4, 4, 0, true, 16}, {1, 6, 4, 0, true, 12},
        {1, 2, 1, 0, false, 16}, {1, 4, 1, 0, false, 16}, {1, 4, 2, 0, false, 12}, {1, 4, 4, 0, false, 12}};
    static const AmtMarchShape pref32[] = {
        {2, 2, 1, 3, true, 16},   // <= 30 levels: 15 cell waves of 2 levels (4096x30x4096: 4.12 against 4.62 ms)
        {2, 3, 1, 0, true, 16}, {2, 4, 1, 0, true, 16}, {2, 3, 2, 0, true, 16}, {2, 4, 2, 0, true, 12}, {2, 4, 2, 0, true, 16},
        {2, 6, 2, 0, true, 12},
        {2, 4, 4, 0, true, 12}, {2, 4, 4, 0, true, 16}, {2, 6, 4, 0, true, 12},
        {2, 4, 1, 0, false, 16}, {2, 4, 2, 0, false, 12}, {2, 4, 2, 0, false, 16},
        {2, 4, 4, 0, false, 12}, {2, 4, 4, 0, false, 16},
        {1, 4, 1, 0, false, 16}, {1, 8, 1, 0, false, 12}, {1, 4, 2, 0, false, 16}, {1, 8, 2, 0, false, 12}, {1, 8, 4, 0, false, 12}};
    const AmtMarchShape *pref = sizeof(T) == 8 ? pref64 : pref32;
    const int npref = sizeof(T) == 8 ? (int)(sizeof pref64 / sizeof pref64[0]) : (int)(sizeof pref32 / sizeof pref32[0]);
    auto usable = [&](const AmtMarchShape &s) {
        if (s.dma && !dma_ok) return false;
        // two columns per lane on an odd row length: the last lane of a row straddles its end; it is never a
        // window lane, its t_1 / v come by DMA chunks, its 2-D column is staged element-wise -- but the
        // register flavour stages rows by lane vectors and needs an even length
        // -- counted from the FIRST TILE'S column 0: tiles that start at an odd window column (rows that are not
        // whole lines, amt_march_col_lo) would leave the row's last column outside every staged pair, and the
        // window's last column then reads a t_1(i+1) nobody wrote (found by the randomised campaign, seed 777
        // case 47199: fp32, 602-element rows, window from column 53)
        if (s.vw > 1 && !s.dma && (p.idim - amt_march_col_lo(p)) % s.vw != 0) return false;
        return amt_march_shape_feasible(wb, s, nk) && amt_march_find<T>(s) != nullptr;
    };
    if (env.kpt || env.hl || env.vw || env.xd >= 0 || env.wm) {
        // forced shape (A/B runs, tests): whatever is left open comes from the preference list, then
        // from the table; the DMA flavour first wherever the layout allows it
        for (int flavour = dma_ok ? 1 : 0; flavour >= 0; --flavour)
            for (int pass = 0; pass < 2; ++pass) {
                int n = 0;
                const AmtMarchEntry<T> *tab = amt_march_table<T>(&n);
                for (int i = 0; i < (pass ? n : npref); ++i) {
                    const AmtMarchShape s = pass ? tab[i].shape : pref[i];
                    if ((int)s.dma != flavour) continue;
                    if (env.kpt && s.kpt != env.kpt) continue;
                    if (env.hl && s.hl != env.hl) continue;
                    if (env.vw && s.vw != env.vw) continue;
                    if (env.xd >= 0 && s.xd != env.xd) continue;
                    if (env.wm && s.wm != env.wm) continue;
                    if (usable(s)) { out = s; return true; }
                }
            }
        return false;                                               // a forced shape that cannot run is an error, not a fallback
    }
    for (int i = 0; i < npref; ++i) {
        if (!usable(pref[i])) continue;
        out = pref[i];
        // Patch-sized launches (WRF patches are a few hundred columns wide): a launch that fits one
        // round of workgroups is latency-bound, and a tile half as wide -- twice the level groups,
        // twice the workgroups -- makes every row of a workgroup cost about 0.8 of the wide one's
        // (profiles/r02_small_domains.md: 64x40x64 11 -> 9 us, 128x60x128 fp64 30 -> 23 us), while in
        // a launch of several rounds it moves fewer bytes per second (512x60x512: 1.17x the time).
        {
            const int nj = p.edges == 1 ? 2 : p.j1 - p.j0 + 1, cus = amt_march_cus(-1);
            auto model = [&](const AmtMarchShape &q) {
                const int tc = (64 / q.hl) * q.vw;
                double c = 0;
                long rounds = 1;
                amt_march_rows((p.i1 - amt_march_col_lo(p)) / tc + 1, nj, amt_march_plan_cus(cus, p.edges), amt_march_max_rows(p), amt_march_rows_cap((int)sizeof(T), q.hl), p.edges == 2 ? amt_march_env().beside_rounds : 1, &c, &rounds);
                return c * (q.hl == out.hl ? 1.0 : rounds == 1 ? 0.8 : 1.2);
            };
            if (out.hl < 4)
                for (int k = i + 1; k < npref; ++k) {
                    const AmtMarchShape &q = pref[k];
                    if (q.vw == out.vw && q.dma == out.dma && q.hl == 2 * out.hl && usable(q)) {
                        if (model(q) < model(out) - 1e-9) out = q;
                        break;
                    }
                }
        }
        return true;
    }
    return false;
}

// What a launch needs beyond the pointers, cached per calling thread for the shape of the call
// (WRF calls this routine every acoustic sub-step with the same bounds: the choice of kernel,
// grid and rows per workgroup is made once).
template <typename T> struct AmtMarchPlan {
    // key
    int nk, idim, kdim, i0, i1, nj, dev, generation, edges;
    bool dma_ok;
    // plan
    bool ok;
    AmtMarchEntry<T> *entry;
    int full, nw;
    int which;             // 2 * cached + full: index of the kernel in the entry
    size_t lds;
    AmtMarchGrid grid;
    char label[360];       // instantiation, rows per workgroup, schedule: what amt_march_last_kernel reports
};

// compute units of device `dev` (-1: the current device); 256 when no device answers
static int amt_march_cus(int dev)
{
    static int cus[64] = {};
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    if (cus[slot] == 0) {
        hipDeviceProp_t prop;
        int n = 256;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        (void)hipGetLastError();
        cus[slot] = n;
    }
    return cus[slot];
}

template <typename T> static bool amt_march_make_plan(const AmtParams<T> &p, AmtMarchPlan<T> &pl)
{
    pl.ok = false;
    AmtMarchShape s;
    if (!amt_march_pick(p, s)) return false;
    const long max_rows = amt_march_max_rows(p);
    if (max_rows < 1) return false;
    const AmtMarchEnv &env = amt_march_env();
    pl.entry = amt_march_find<T>(s);
    const int lw = s.kpt * s.hl, tc = (64 / s.hl) * s.vw;
    pl.full = (p.nk % lw == 0 && amt_march_full_build_ok((int)sizeof(T), s)) ? 1 : 0;
    // rows that are not whole 128-byte lines share their edge lines between neighbouring tiles: no nt policy there
    const bool cached = env.nt < 0 ? ((size_t)p.idim * sizeof(T)) % 128 != 0 : env.nt == 0;
    pl.which = (cached ? 2 : 0) + pl.full;
    pl.nw = amt_march_waves(s, p.nk);
    pl.lds = amt_march_lds_bytes((int)sizeof(T), s, p.nk);
    AmtMarchGrid &g = pl.grid;
    const int nj = p.j1 - p.j0 + 1;
    g.col_lo = amt_march_col_lo(p);
    g.ntile_i = (p.i1 - g.col_lo) / tc + 1;
    int jrows = env.jrows;
    if (jrows < 1) jrows = amt_march_rows(g.ntile_i, nj, amt_march_plan_cus(amt_march_cus(pl.dev), p.edges), max_rows, amt_march_rows_cap((int)sizeof(T), s.hl), p.edges == 2 ? amt_march_env().beside_rounds : 1, nullptr, nullptr);
    if (jrows > nj) jrows = nj;
    if (jrows > max_rows) jrows = (int)max_rows;
    g.jrows = jrows;
    g.jstep = jrows;
    g.njblk = (nj + jrows - 1) / jrows;
    if (p.edges == 1) {                              // rows j0 and j1 only: two one-row blocks
        g.jrows = 1;
        g.jstep = nj - 1;
        g.njblk = 2;
    }
    g.nwg = g.ntile_i * g.njblk;
    g.xchunk = env.xchunk > 0 ? env.xchunk : 0;
    if (pl.lds > 64 * 1024) {
        // the attribute is per device and per kernel instantiation: allow all of the CU's LDS once
        // (what a launch occupies is its own dynamic size, not this ceiling); one-shot calls plan from
        // several host threads at once
        static std::mutex grant;
        std::lock_guard<std::mutex> lk(grant);
        if (!(pl.entry->lds_granted[pl.which] >> (pl.dev & 31) & 1u)) {
            hipError_t e = hipFuncSetAttribute(pl.entry->kernel[pl.which], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { (void)hipGetLastError(); return false; }
            pl.entry->lds_granted[pl.which] |= 1u << (pl.dev & 31);
        }
    }
    if (env.verbose)
        fprintf(stderr, "[amt march] nk %d idim %d window i %d..%d, %d rows -> %s %s: %d waves, %zu B LDS, %d tiles x %d blocks of %d rows\n",
                p.nk, p.idim, p.i0, p.i1, nj, cached ? pl.entry->name_cached : pl.entry->name, pl.full ? "FULL" : "ragged", pl.nw, pl.lds, g.ntile_i, g.njblk, g.jrows);
    snprintf(pl.label, sizeof pl.label, "%s %s jrows=%d", cached ? pl.entry->name_cached : pl.entry->name, pl.full ? "FULL" : "ragged", g.jrows);
    snprintf(g_march_last, sizeof g_march_last, "%s", pl.label);
    pl.ok = true;
    return true;
}

template <typename T> static const AmtMarchPlan<T> *amt_march_plan(const AmtParams<T> &p)
{
    constexpr int NSLOT = 8;
    static thread_local AmtMarchPlan<T> cache[NSLOT];
    static thread_local int used = 0, next = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const int nj = p.j1 - p.j0 + 1;
    const bool dma_ok = amt_march_dma_layout_ok(p);
    for (int i = 0; i < used; ++i) {
        const AmtMarchPlan<T> &c = cache[i];
        if (c.nk == p.nk && c.idim == p.idim && c.kdim == p.kdim && c.i0 == p.i0 && c.i1 == p.i1 && c.nj == nj
            && c.dev == dev && c.dma_ok == dma_ok && c.generation == g_march_generation && c.edges == p.edges)
            return c.ok ? &c : nullptr;
    }
    const int slot = used < NSLOT ? used++ : (next = (next + 1) % NSLOT);
    AmtMarchPlan<T> &c = cache[slot];
    c.nk = p.nk; c.idim = p.idim; c.kdim = p.kdim; c.i0 = p.i0; c.i1 = p.i1; c.nj = nj; c.dev = dev; c.dma_ok = dma_ok;
    c.generation = g_march_generation;
    c.edges = p.edges;
    amt_march_make_plan(p, c);
    return c.ok ? &c : nullptr;
}

template <typename T> bool amt_march_supported(const AmtParams<T> &p)
{
    if (p.i1 < p.i0 || p.j1 < p.j0) return true;
    return amt_march_plan<T>(p) != nullptr;
}

template <typename T>
hipError_t amt_launch_march(hipStream_t stream, const AmtParams<T> &p)
{
    if (p.i1 < p.i0 || p.j1 < p.j0) return hipSuccess;
    const AmtMarchPlan<T> *pl = amt_march_plan<T>(p);
    if (!pl) return hipErrorNotSupported;
    pl->entry->launch[pl->which](stream, p, pl->grid, pl->nw, pl->lds);
    // the kernel the calling thread LAUNCHED last (a slab's one-row edge launch does not rename its interior kernel)
    if (p.edges != 1 && strcmp(g_march_last, pl->label) != 0) snprintf(g_march_last, sizeof g_march_last, "%s", pl->label);
    return hipGetLastError();
}

// The shapes amt_march_pick can select without an AMT_MARCH_* override, as kernel names (the
// build-time scratch check of tools/kernel_resources.py and tests/test_kernel_resources.py):
// one name per line into buf; returns the number of bytes needed.
extern "C" int amt_march_selectable(char *buf, int cap)
{
    std::string s;
    auto add = [&](const char *t, const AmtMarchShape &q) {
        char line[160];
        for (int full = 0; full < (amt_march_full_build_ok(t[0] == 'd' ? 8 : 4, q) ? 2 : 1); ++full) {
            for (int ntl : {AMT_NT_LOAD, 0}) {                     // the two cache policies of the once-read streams
                snprintf(line, sizeof line, "amt_march_kernel<%s, %d, %d, %d, %d, %s, %s, %d, %d>\n", t, q.vw, q.kpt, q.hl, q.xd,
                         full ? "true" : "false", q.dma ? "true" : "false", q.wm, ntl);
                if (s.find(line) == std::string::npos) s += line;
            }
        }
    };
    for (int nk = 1; nk <= 400; ++nk)
        for (int lay = 0; lay < 2; ++lay)                            // even and odd row lengths
            for (int win = 0; win < 2; ++win) {                      // a whole domain and a patch-sized launch
                AmtParams<double> pd = {};
                AmtParams<float> pf = {};
                pd.nk = pf.nk = nk;
                pd.kdim = pf.kdim = nk + 1;                          // kme = kte: one level row behind the window's last
                pd.idim = pf.idim = lay ? 4099 : 4160;
                pd.i0 = pf.i0 = 32;
                pd.i1 = pf.i1 = win ? 95 : 4095;
                pd.j0 = pf.j0 = 1;
                pd.j1 = pf.j1 = win ? 64 : 4096;
                AmtMarchShape q;
                if (amt_march_pick(pd, q)) add("double", q);
                if (amt_march_pick(pf, q)) add("float", q);
            }
    if (buf && cap > 0) {
        const int n = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size() + 1;
}

template bool amt_march_supported<float>(const AmtParams<float> &);
template bool amt_march_supported<double>(const AmtParams<double> &);
template hipError_t amt_launch_march<float>(hipStream_t, const AmtParams<float> &);
template hipError_t amt_launch_march<double>(hipStream_t, const AmtParams<double> &);
