/*
 * advance_mu_t_driver.c -- C99 host driver of the MI355X advance_mu_t path.
 *
 * The reference ships a C driver next to its Fortran one (advance_mu_t_driver.c:37-257: read the dimensions and the 26
 * arrays, time one call with gettimeofday, compare the 8 outputs).  This is that flow on top of the C-ABI
 * (include/amt_advance_mu_t.h): the inputs are the seeded synthetic fields of include/amt_synth.h (the reference's
 * /data2/... dump is not shipped), the call is the one-shot drop-in amt_advance_mu_t_f32 / _f64 -- the reference's C
 * signature (advance_mu_t.h:10-23) with the config struct spelled out as its three flags and without kds -- and the
 * outputs are dumped as raw native-endian streams for an external checker (tests/test_gpu_21_fortran_host.py holds them
 * against the oracle).  Built twice: -DAMT_REAL=float (the reference's C version is float only) and -DAMT_REAL=double.
 *
 *   advance_mu_t_c_driver_f32|_f64 [NI NK NJ [nsweeps [outdir [flags]]]]      flags: 0 none, 1 specified, 2 nested,
 *                                                                             3 specified + periodic_x
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "amt_advance_mu_t.h"
#include "amt_synth.h"

#ifndef AMT_REAL
#define AMT_REAL float
#endif
typedef AMT_REAL real;

static void *xmalloc(size_t n)
{
    void *p = malloc(n);
    if (!p) { fprintf(stderr, "out of memory (%zu bytes)\n", n); exit(1); }
    return p;
}

static void dump(const char *dir, const char *name, const real *a, size_t n)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(a, sizeof(real), n, f) != n) { fprintf(stderr, "cannot write %s\n", path); exit(1); }
    fclose(f);
}

int main(int argc, char **argv)
{
    int ni = 64, nk = 40, nj = 64, nsweeps = 3, iflag = 0;              /* BASELINE.json configs[0] */
    const char *outdir = "";
    if (argc > 3) { ni = atoi(argv[1]); nk = atoi(argv[2]); nj = atoi(argv[3]); }
    if (argc > 4) nsweeps = atoi(argv[4]);
    if (argc > 5) outdir = argv[5];
    if (argc > 6) iflag = atoi(argv[6]);
    const int specified = (iflag == 1 || iflag == 3), nested = (iflag == 2), periodic_x = (iflag == 3);

    /* single-patch domain, SURVEY.md section 8 convention (the bounds the reference reads from ids.bin ... kte.bin) */
    const int ids = 1, ide = ni + 1, jds = 1, jde = nj + 1, kde = nk + 1;
    const int ims = 0, ime = ni + 1, jms = 0, jme = nj + 1, kms = 1, kme = nk + 1;
    const int its = 1, ite = ide, jts = 1, jte = jde, kts = 1, kte = kde;
    const long idim = ime - ims + 1, kdim = kme - kms + 1, jdim = jme - jms + 1;
    const size_t n3 = (size_t)idim * kdim * jdim, n2 = (size_t)idim * jdim, n1 = (size_t)kdim;

    real *a[AMT_F_COUNT];
    for (int f = 0; f < AMT_F_COUNT; ++f) {
        const int rank = amt_field_rank(f);
        const size_t n = rank == 3 ? n3 : rank == 2 ? n2 : n1;
        a[f] = (real *)xmalloc(n * sizeof(real));
        const int rc = amt_synth_fill_host(f, (int)sizeof(real), a[f], 12345u,
                                           rank == 1 ? 1 : idim, rank == 2 ? 1 : kdim, rank == 1 ? 1 : jdim,
                                           rank == 1 ? 0 : ims, rank == 2 ? 0 : kms - 1, rank == 1 ? 0 : jms,
                                           ni + 2, nk + 1, nj + 2);
        if (rc != AMT_OK) { fprintf(stderr, "amt_synth_fill_host: %s\n", amt_last_error()); return 1; }
    }
    printf("advance_mu_t (C host) %dx%dx%d real*%d  HIP devices: %d\n", ni, nk, nj, (int)sizeof(real), amt_device_count());

    struct timeval t0, t1;
    gettimeofday(&t0, NULL);                                            /* advance_mu_t_driver.c:222-225 */
    for (int s = 0; s < nsweeps; ++s) {
#define A(f) a[AMT_F_##f]
        const int rc =
#if defined(AMT_REAL_IS_DOUBLE)
            amt_advance_mu_t_f64(
#else
            amt_advance_mu_t_f32(
#endif
                A(WW), A(WW_1), A(U), A(U_1), A(V), A(V_1), A(MU), A(MUT), A(MUAVE), A(MUTS), A(MUU), A(MUV), A(MUDF),
                A(T), A(T_1), A(T_AVE), A(FT), A(MU_TEND),
                (real)AMT_SYNTH_RDX, (real)AMT_SYNTH_RDY, (real)AMT_SYNTH_DTS, (real)AMT_SYNTH_EPSSM,
                A(DNW), A(FNM), A(FNP), A(RDNW), A(MSFUY), A(MSFVX_INV), A(MSFTX), A(MSFTY),
                periodic_x, specified, nested,
                ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte);
#undef A
        if (rc != AMT_OK) {                                             /* the reference prints and exit(1)s */
            fprintf(stderr, "advance_mu_t: status %d (%s): %s\n", rc, amt_status_string(rc), amt_last_error());
            return 1;
        }
    }
    gettimeofday(&t1, NULL);
    const double ms = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) * 1e-3) / (nsweeps > 0 ? nsweeps : 1);
    printf("one-shot host path: %d calls, %.4f ms/call (H2D + kernel + D2H), %.1f Mcells/s\n", nsweeps, ms,
           (double)ni * nk * nj / (ms * 1e-3) / 1e6);
    if (outdir[0]) {
        dump(outdir, "ww", a[AMT_F_WW], n3); dump(outdir, "t", a[AMT_F_T], n3); dump(outdir, "t_ave", a[AMT_F_T_AVE], n3);
        dump(outdir, "mu", a[AMT_F_MU], n2); dump(outdir, "muave", a[AMT_F_MUAVE], n2);
        dump(outdir, "muts", a[AMT_F_MUTS], n2); dump(outdir, "mudf", a[AMT_F_MUDF], n2);
    }
    amt_host_release();
    for (int f = 0; f < AMT_F_COUNT; ++f) free(a[f]);
    return 0;
}
