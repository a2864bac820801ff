/*
 * oracle/oracle_bench.c
 *
 * TEST INFRASTRUCTURE ONLY: the timing harness of bench.py's cpu_baseline leg (SURVEY.md
 * section 8(d)).  The same C restatement of the reference Fortran as
 * advance_mu_t_oracle.c (the shared advance_mu_t_oracle_impl.h), built -O3 -march=native for
 * the host it runs on -- still without FMA contraction, so its results are the checker's bits
 * (tests/test_oracle_golden.py compares the two builds) -- plus a driver that does what a
 * careful CPU run of WRF does:
 *
 *   - the 26 arrays of an NI x NK x NJ domain are allocated untouched and FIRST TOUCHED by the
 *     thread that will later compute those rows (static j-tiles, the scheme sketched in the
 *     reference driver, advance_mu_t_driver.f90:175-209), so every tile's pages sit on its own
 *     NUMA node -- r01's numpy-filled arrays lived on one node and the 256-thread whole-domain
 *     run was bound by that;
 *   - inputs are the seeded generator of include/amt_synth.h (the GPU run's inputs);
 *   - `reps` sweeps, each a j-tiled call of the routine, timed one by one.
 *
 * Nothing of the product links or loads this file.
 */
#include <omp.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/amt_synth.h"

#define ORACLE_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORACLE_MAX(a, b) ((a) > (b) ? (a) : (b))

#define REAL float
#define ORACLE_FN(name) name##_f32
#include "advance_mu_t_oracle_impl.h"
#undef REAL
#undef ORACLE_FN

#define REAL double
#define ORACLE_FN(name) name##_f64
#include "advance_mu_t_oracle_impl.h"
#undef REAL
#undef ORACLE_FN

/*
 * dtype_bytes 4 or 8; the domain uses the minimal single-patch memory (0:NI+1, 1:NK+1, 0:NJ+1)
 * and is treated as a j-slab [gj0, gj0+NJ+1] of a domain of gnj rows (the generator is indexed
 * globally).  ms[reps] receives the wall time of each sweep; *checksum a sum over mu (keeps the
 * work observable).  Returns 0, or 1 when the allocation fails, 2 on a bad argument.
 */
/* `fn` (oracle_bench_fn only): NULL = the C restatement, j-tiled here; otherwise a routine with the
 * argument list of oracle_advance_mu_t_omp_f32/_f64 (arrays, scalars, flags, bounds, nthreads) that tiles
 * j itself -- the Fortran CPU path, oracle/fortran/advance_mu_t_cpu.f90 (amt_cpu_fortran).  For that
 * case this file is built with the LLVM OpenMP runtime the Fortran uses (liboracle_bench_llvm.so), so
 * that the threads that first touch a tile's pages are the threads that compute it. */
typedef int (*oracle_fn_f32)(float *, const float *, const float *, const float *, const float *, const float *,
                             float *, const float *, float *, float *, const float *, const float *, float *, float *,
                             const float *, float *, const float *, const float *, float, float, float, float,
                             const float *, const float *, const float *, const float *, const float *, const float *,
                             const float *, const float *, int, int, int, int, int, int, int, int, int, int, int, int,
                             int, int, int, int, int, int, int, int, int);
typedef int (*oracle_fn_f64)(double *, const double *, const double *, const double *, const double *, const double *,
                             double *, const double *, double *, double *, const double *, const double *, double *, double *,
                             const double *, double *, const double *, const double *, double, double, double, double,
                             const double *, const double *, const double *, const double *, const double *, const double *,
                             const double *, const double *, int, int, int, int, int, int, int, int, int, int, int, int,
                             int, int, int, int, int, int, int, int, int);

int oracle_bench_fn(int dtype_bytes, int ni, int nk, int nj, long gj0, long gnj, uint64_t seed,
                    int nthreads, int reps, double *ms, double *checksum, double *fill_seconds, void *fn)
{
    if ((dtype_bytes != 4 && dtype_bytes != 8) || ni < 1 || nk < 1 || nj < 1 || reps < 1 || !ms) return 2;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nj) nthreads = nj;
    const long idim = ni + 2, kdim = nk + 1, jdim = nj + 2;
    const size_t W = (size_t)dtype_bytes;
    void *a[AMT_F_COUNT];
    for (int f = 0; f < AMT_F_COUNT; ++f) {
        const int r = amt_field_rank(f);
        const size_t n = r == 3 ? (size_t)idim * kdim * jdim : r == 2 ? (size_t)idim * jdim : (size_t)kdim;
        if (posix_memalign(&a[f], 4096, n * W + 4096)) {
            for (int g = 0; g < f; ++g) free(a[g]);
            return 1;
        }
    }
    /* a one-thread run has no placement to get right: its fill may use more threads (ORACLE_BENCH_FILL_THREADS) */
    int fillthreads = nthreads;
    if (nthreads == 1 && getenv("ORACLE_BENCH_FILL_THREADS")) {
        fillthreads = atoi(getenv("ORACLE_BENCH_FILL_THREADS"));
        if (fillthreads < 1) fillthreads = 1;
        if (fillthreads > nj) fillthreads = nj;
    }
    const double t_fill = omp_get_wtime();
    /* first touch: thread id fills memory rows of its compute tile (plus the halo rows at the
       two ends); rows are jm = 0 .. jdim-1, compute rows j = 1 .. NJ */
#pragma omp parallel for num_threads(fillthreads) schedule(static, 1)
    for (int id = 0; id < fillthreads; ++id) {
        long lo = 1 + ((long)nj * id) / fillthreads, hi = 1 + ((long)nj * (id + 1)) / fillthreads - 1;
        if (id == 0) lo = 0;
        if (id == fillthreads - 1) hi = jdim - 1;
        for (int f = 0; f < AMT_F_COUNT; ++f) {
            const int r = amt_field_rank(f);
            if (r == 1) {
                if (id != 0) continue;
                for (long k = 0; k < kdim; ++k) {
                    const double x = amt_synth_value(f, seed, 0, k, 0, idim, kdim, gnj + 2);
                    if (W == 8) ((double *)a[f])[k] = x; else ((float *)a[f])[k] = (float)x;
                }
                continue;
            }
            const long kd = r == 3 ? kdim : 1;
            for (long j = lo; j <= hi; ++j)
                for (long k = 0; k < kd; ++k) {
                    const size_t row = ((size_t)j * kd + k) * idim;
                    for (long i = 0; i < idim; ++i) {
                        const double x = amt_synth_value(f, seed, i, k, gj0 + j, idim, kdim, gnj + 2);
                        if (W == 8) ((double *)a[f])[row + i] = x; else ((float *)a[f])[row + i] = (float)x;
                    }
                }
        }
    }
    if (fill_seconds) *fill_seconds = omp_get_wtime() - t_fill;

#define ARGS(T)                                                                                          \
            (T *)a[AMT_F_WW], (T *)a[AMT_F_WW_1], (T *)a[AMT_F_U], (T *)a[AMT_F_U_1], (T *)a[AMT_F_V],  \
            (T *)a[AMT_F_V_1], (T *)a[AMT_F_MU], (T *)a[AMT_F_MUT], (T *)a[AMT_F_MUAVE], (T *)a[AMT_F_MUTS], \
            (T *)a[AMT_F_MUU], (T *)a[AMT_F_MUV], (T *)a[AMT_F_MUDF], (T *)a[AMT_F_T], (T *)a[AMT_F_T_1], \
            (T *)a[AMT_F_T_AVE], (T *)a[AMT_F_FT], (T *)a[AMT_F_MU_TEND],                                \
            (T)AMT_SYNTH_RDX, (T)AMT_SYNTH_RDY, (T)AMT_SYNTH_DTS, (T)AMT_SYNTH_EPSSM,                    \
            (T *)a[AMT_F_DNW], (T *)a[AMT_F_FNM], (T *)a[AMT_F_FNP], (T *)a[AMT_F_RDNW],                 \
            (T *)a[AMT_F_MSFUY], (T *)a[AMT_F_MSFVX_INV], (T *)a[AMT_F_MSFTX], (T *)a[AMT_F_MSFTY],      \
            0, 0, 0, 1, ni + 1, 1, nj + 1, nk + 1, 0, ni + 1, 0, nj + 1, 1, nk + 1,                      \
            1, ni + 1, lo, hi, 1, nk + 1
    for (int rep = 0; rep < reps; ++rep) {
        const double t0 = omp_get_wtime();
        if (fn) {
            const int lo = 1, hi = nj;
            int rc;
            if (W == 8) rc = ((oracle_fn_f64)fn)(ARGS(double), nthreads);
            else rc = ((oracle_fn_f32)fn)(ARGS(float), nthreads);
            if (rc) { for (int f = 0; f < AMT_F_COUNT; ++f) free(a[f]); return 3; }
            ms[rep] = (omp_get_wtime() - t0) * 1e3;
            continue;
        }
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
        for (int id = 0; id < nthreads; ++id) {
            const int lo = 1 + (int)(((long)nj * id) / nthreads), hi = 1 + (int)(((long)nj * (id + 1)) / nthreads) - 1;
            if (hi < lo) continue;
            if (W == 8) (void)oracle_advance_mu_t_f64(ARGS(double));
            else (void)oracle_advance_mu_t_f32(ARGS(float));
#undef ARGS
        }
        ms[rep] = (omp_get_wtime() - t0) * 1e3;
    }
    double s = 0.0;
    for (long e = 0; e < idim * jdim; ++e) s += W == 8 ? ((double *)a[AMT_F_MU])[e] : (double)((float *)a[AMT_F_MU])[e];
    if (checksum) *checksum = s;
    for (int f = 0; f < AMT_F_COUNT; ++f) free(a[f]);
    return 0;
}

int oracle_bench(int dtype_bytes, int ni, int nk, int nj, long gj0, long gnj, uint64_t seed,
                 int nthreads, int reps, double *ms, double *checksum, double *fill_seconds)
{
    return oracle_bench_fn(dtype_bytes, ni, nk, nj, gj0, gnj, seed, nthreads, reps, ms, checksum, fill_seconds, NULL);
}
