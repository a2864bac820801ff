// amt_params.h -- kernel argument block and the loop-bound logic of advance_mu_t.
#pragma once
#include <stddef.h>

// All array arguments of SUBROUTINE advance_mu_t (module_small_step_em.f90:7-14)
// as device pointers to element (ims,kms,jms) / (ims,jms) / (kms), plus the
// rebased (zero-based, memory-relative) compute window.
template <typename T>
struct AmtParams {
    // INOUT / OUT
    T *ww, *mu, *muave, *muts, *mudf, *t, *t_ave;
    // IN
    const T *ww_1, *u, *u_1, *v, *v_1, *mut, *muu, *muv, *t_1, *ft, *mu_tend;
    const T *dnw, *fnm, *fnp, *rdnw, *msfuy, *msfvx_inv, *msftx, *msfty;
    T rdx, rdy, dts, epssm;
    // memory extents (elements)
    int idim, kdim;
    long jstride;          // idim * kdim
    // compute window, zero-based memory indices, inclusive
    int i0, i1;            // i_start - ims .. i_end - ims
    int j0, j1;            // j_start - jms .. j_end - jms
    int k1;                // memory index of Fortran level k = 1   (1 - kms)
    int nk;                // k_end = kte - 1 : levels 1..nk are updated
    int edges;             // 1: only rows j0 and j1 of the window (a j-slab's two boundary rows in ONE launch)
                           // 2: the whole window, launched BESIDE another stream's kernels (a j-slab's interior rows while its
                           //    halo exchange runs): planned in at least two rounds of workgroups, so that the other stream's
                           //    kernels get compute units at a round boundary instead of behind the whole launch
};

// Compute window, module_small_step_em.f90:91-106.
struct AmtWindow {
    int i_start, i_end, j_start, j_end, k_start, k_end;
};

static inline AmtWindow amt_window(int periodic_x, int specified, int nested,
                                   int ids, int ide, int jds, int jde,
                                   int its, int ite, int jts, int jte, int kts, int kte)
{
    AmtWindow w;
    w.i_start = its;
    w.i_end   = ite < ide - 1 ? ite : ide - 1;
    w.j_start = jts;
    w.j_end   = jte < jde - 1 ? jte : jde - 1;
    w.k_start = kts;
    w.k_end   = kte - 1;
    if (!periodic_x) {
        if (specified || nested) {
            w.i_start = its > ids + 1 ? its : ids + 1;
            w.i_end   = ite < ide - 2 ? ite : ide - 2;
        }
    }
    if (specified || nested) {
        w.j_start = jts > jds + 1 ? jts : jds + 1;
        w.j_end   = jte < jde - 2 ? jte : jde - 2;
    }
    return w;
}
