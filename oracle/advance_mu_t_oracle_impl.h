/*
 * oracle/advance_mu_t_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see advance_mu_t_oracle.c).
 *
 * Body of the CPU restatement, included once per precision with
 *   REAL   = float | double
 *   ORACLE_FN(name) -> name##_f32 | name##_f64
 *
 * Restates /root/reference/module_small_step_em.f90:7-252 (SUBROUTINE advance_mu_t)
 * loop for loop and operation for operation: the three phases are NOT fused, the
 * scratch arrays wdtn/dvdxi/dmdt are tile-local exactly like the Fortran stack
 * arrays (module_small_step_em.f90:74-75), every expression keeps the Fortran
 * parenthesisation and left-to-right order, and the file is compiled with
 * -ffp-contract=off so that no FMA is formed.  The debug dumps of
 * module_small_step_em.f90:175-189 are a side effect of the sample, not of WRF,
 * and are not reproduced.
 *
 * Bounds are Fortran-style inclusive and passed unchanged; indexing macros below
 * rebase them, i fastest: element (i,k,j) lives at
 *   ((j-jms)*kdim + (k-kms))*idim + (i-ims)       (advance_mu_t.c:8-9 has the same
 * map after its own index normalisation, advance_mu_t.c:33-55).
 */

#define A3(a, i, k, j) (a)[((size_t)((j) - jms) * kdim + (size_t)((k) - kms)) * idim + (size_t)((i) - ims)]
#define A2(a, i, j)    (a)[(size_t)((j) - jms) * idim + (size_t)((i) - ims)]
#define A1(a, k)       (a)[(k) - kms]
/* tile-local scratch (its:ite, kts:kmax) and (its:ite) */
#define L2D(a, i, k)   (a)[(size_t)((k) - kts) * ni_t + (size_t)((i) - its)]
#define L1D(a, i)      (a)[(i) - its]

int ORACLE_FN(oracle_advance_mu_t)(
    REAL *ww, const REAL *ww_1, const REAL *u, const REAL *u_1,
    const REAL *v, const REAL *v_1,
    REAL *mu, const REAL *mut, REAL *muave, REAL *muts,
    const REAL *muu, const REAL *muv,
    REAL *mudf, REAL *t, const REAL *t_1,
    REAL *t_ave, const REAL *ft, const REAL *mu_tend,
    REAL rdx, REAL rdy, REAL dts, REAL epssm,
    const REAL *dnw, const REAL *fnm, const REAL *fnp, const REAL *rdnw,
    const REAL *msfuy, const REAL *msfvx_inv,
    const REAL *msftx, const REAL *msfty,
    int periodic_x, int specified, int nested,
    int ids, int ide, int jds, int jde, int kde,
    int ims, int ime, int jms, int jme, int kms, int kme,
    int its, int ite, int jts, int jte, int kts, int kte)
{
    const size_t idim = (size_t)(ime - ims + 1);
    const size_t kdim = (size_t)(kme - kms + 1);
    (void)jme;

    int i, j, k;
    int i_start, i_end, j_start, j_end, k_start, k_end;

    /* module_small_step_em.f90:91-106 -- loop bounds from domain/tile/BC flags */
    i_start = its;
    i_end   = ORACLE_MIN(ite, ide - 1);
    j_start = jts;
    j_end   = ORACLE_MIN(jte, jde - 1);
    k_start = kts;
    k_end   = kte - 1;
    if (!periodic_x) {
        if (specified || nested) {
            i_start = ORACLE_MAX(its, ids + 1);
            i_end   = ORACLE_MIN(ite, ide - 2);
        }
    }
    if (specified || nested) {
        j_start = ORACLE_MAX(jts, jds + 1);
        j_end   = ORACLE_MIN(jte, jde - 2);
    }

    /* module_small_step_em.f90:74-75 -- REAL, DIMENSION(its:ite,kts:kte) :: wdtn, dvdxi ;
     * (its:ite) :: dmdt.  wdtn(i,kde) is written at :221, so the Fortran is only
     * defined for kde <= kte; the restatement refuses anything else. */
    if (kde > kte || kde < kts || kts > 1) return 2;   /* :159,:168,:220 use literal k=1,2 */
    if (ite < its || kte < kts) return 0;
    const size_t ni_t = (size_t)(ite - its + 1);
    const size_t nk_t = (size_t)(kte - kts + 1);
    REAL *wdtn  = (REAL *)malloc(ni_t * nk_t * sizeof(REAL));
    REAL *dvdxi = (REAL *)malloc(ni_t * nk_t * sizeof(REAL));
    REAL *dmdt  = (REAL *)malloc(ni_t * sizeof(REAL));
    if (!wdtn || !dvdxi || !dmdt) { free(wdtn); free(dvdxi); free(dmdt); return 1; }

    /* ---- CALCULATION OF WW (dETA/dt): module_small_step_em.f90:112-174 ---- */
    for (j = j_start; j <= j_end; j++) {

        for (i = i_start; i <= i_end; i++)            /* :114-116 */
            L1D(dmdt, i) = (REAL)0.;

        for (k = k_start; k <= k_end; k++) {          /* :140-149 */
            for (i = i_start; i <= i_end; i++) {
                L2D(dvdxi, i, k) = A2(msftx, i, j) * A2(msfty, i, j) * (
                      rdy * ( (A3(v, i, k, j + 1) + A2(muv, i, j + 1) * A3(v_1, i, k, j + 1) * A2(msfvx_inv, i, j + 1))
                            - (A3(v, i, k, j    ) + A2(muv, i, j    ) * A3(v_1, i, k, j    ) * A2(msfvx_inv, i, j    )) )
                    + rdx * ( (A3(u, i + 1, k, j) + A2(muu, i + 1, j) * A3(u_1, i + 1, k, j) / A2(msfuy, i + 1, j))
                            - (A3(u, i    , k, j) + A2(muu, i    , j) * A3(u_1, i    , k, j) / A2(msfuy, i    , j)) ));
                L1D(dmdt, i) = L1D(dmdt, i) + A1(dnw, k) * L2D(dvdxi, i, k);
            }
        }

        for (i = i_start; i <= i_end; i++) {          /* :151-157 */
            A2(muave, i, j) = A2(mu, i, j);
            A2(mu, i, j)    = A2(mu, i, j) + dts * (L1D(dmdt, i) + A2(mu_tend, i, j));
            A2(mudf, i, j)  = (L1D(dmdt, i) + A2(mu_tend, i, j));
            A2(muts, i, j)  = A2(mut, i, j) + A2(mu, i, j);
            A2(muave, i, j) = (REAL).5 * (((REAL)1. + epssm) * A2(mu, i, j) + ((REAL)1. - epssm) * A2(muave, i, j));
        }

        for (k = 2; k <= k_end; k++) {                /* :159-163 (k literally starts at 2) */
            for (i = i_start; i <= i_end; i++) {
                A3(ww, i, k, j) = A3(ww, i, k - 1, j)
                    - A1(dnw, k - 1) * (L1D(dmdt, i) + L2D(dvdxi, i, k - 1) + A2(mu_tend, i, j)) / A2(msfty, i, j);
            }
        }

        for (k = 1; k <= k_end; k++) {                /* :168-172 (k literally starts at 1) */
            for (i = i_start; i <= i_end; i++)
                A3(ww, i, k, j) = A3(ww, i, k, j) - A3(ww_1, i, k, j);
        }
    }

    /* ---- CALCULATION OF THETA, pre-update: module_small_step_em.f90:208-215 ---- */
    for (j = j_start; j <= j_end; j++) {
        for (k = 1; k <= k_end; k++) {
            for (i = i_start; i <= i_end; i++) {
                A3(t_ave, i, k, j) = A3(t, i, k, j);
                A3(t, i, k, j)     = A3(t, i, k, j) + A2(msfty, i, j) * dts * A3(ft, i, k, j);
            }
        }
    }

    /* ---- vertical flux + flux-form theta update: module_small_step_em.f90:217-250 ---- */
    for (j = j_start; j <= j_end; j++) {

        for (i = i_start; i <= i_end; i++) {          /* :219-222 */
            L2D(wdtn, i, 1)   = (REAL)0.;
            L2D(wdtn, i, kde) = (REAL)0.;
        }

        for (k = 2; k <= k_end; k++) {                /* :224-229 */
            for (i = i_start; i <= i_end; i++)
                L2D(wdtn, i, k) = A3(ww, i, k, j) * (A1(fnm, k) * A3(t_1, i, k, j) + A1(fnp, k) * A3(t_1, i, k - 1, j));
        }

        for (k = 1; k <= k_end; k++) {                /* :234-248 */
            for (i = i_start; i <= i_end; i++) {
                A3(t, i, k, j) = A3(t, i, k, j) - dts * A2(msfty, i, j) * (
                        A2(msftx, i, j) * (
                            (REAL).5 * rdy *
                              ( A3(v, i, k, j + 1) * (A3(t_1, i, k, j + 1) + A3(t_1, i, k, j    ))
                              - A3(v, i, k, j    ) * (A3(t_1, i, k, j    ) + A3(t_1, i, k, j - 1)) )
                          + (REAL).5 * rdx *
                              ( A3(u, i + 1, k, j) * (A3(t_1, i + 1, k, j) + A3(t_1, i    , k, j))
                              - A3(u, i    , k, j) * (A3(t_1, i    , k, j) + A3(t_1, i - 1, k, j)) ) )
                      + A1(rdnw, k) * (L2D(wdtn, i, k + 1) - L2D(wdtn, i, k)) );
            }
        }
    }

    free(wdtn);
    free(dvdxi);
    free(dmdt);
    return 0;
}

/*
 * j-tiled OpenMP wrapper: every thread calls the routine above on its own
 * (jts:jte) sub-tile -- the scheme sketched (commented out) in the reference
 * driver, advance_mu_t_driver.f90:175-209.  Columns never read another column's
 * outputs, so the result is bit-identical to one call.
 * Used as the timed CPU baseline in bench.py.
 */
int ORACLE_FN(oracle_advance_mu_t_omp)(
    REAL *ww, const REAL *ww_1, const REAL *u, const REAL *u_1,
    const REAL *v, const REAL *v_1,
    REAL *mu, const REAL *mut, REAL *muave, REAL *muts,
    const REAL *muu, const REAL *muv,
    REAL *mudf, REAL *t, const REAL *t_1,
    REAL *t_ave, const REAL *ft, const REAL *mu_tend,
    REAL rdx, REAL rdy, REAL dts, REAL epssm,
    const REAL *dnw, const REAL *fnm, const REAL *fnp, const REAL *rdnw,
    const REAL *msfuy, const REAL *msfvx_inv,
    const REAL *msftx, const REAL *msfty,
    int periodic_x, int specified, int nested,
    int ids, int ide, int jds, int jde, int kde,
    int ims, int ime, int jms, int jme, int kms, int kme,
    int its, int ite, int jts, int jte, int kts, int kte,
    int nthreads)
{
    int rc = 0;
    const int nj = jte - jts + 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nj) nthreads = nj > 0 ? nj : 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int id = 0; id < nthreads; id++) {
        const int lo = jts + (int)(((long)nj * id) / nthreads);
        const int hi = jts + (int)(((long)nj * (id + 1)) / nthreads) - 1;
        if (hi < lo) continue;
        int r = ORACLE_FN(oracle_advance_mu_t)(
            ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,
            t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,
            msfuy, msfvx_inv, msftx, msfty, periodic_x, specified, nested,
            ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,
            its, ite, lo, hi, kts, kte);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

#undef A3
#undef A2
#undef A1
#undef L2D
#undef L1D
