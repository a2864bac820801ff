/*
 * include/amt_synth.h -- closed-form, seeded synthetic WRF-shaped inputs.
 *
 * The reference reads its inputs from big-endian dumps of a real WRF V3.4.1 run
 * (advance_mu_t_driver.f90:38-167, advance_mu_t_driver.c:60-219) that are not
 * shipped.  This header is the stand-in: every element of every argument array
 * of advance_mu_t is a pure function of (field, seed, GLOBAL i,k,j index), so
 *   - the host (tests, Fortran driver) and a device fill kernel produce
 *     identical bits (only exactly-rounded IEEE +,*,/ on values built from
 *     integers; no libm; compile with -ffp-contract=off),
 *   - a j-slab owner can fill its slab and halo rows without communication,
 *   - full-size (4096x60x4096) validation can regenerate any j-slab on the host.
 * Value ranges follow SURVEY.md section 8(d): divisors msfuy/msfty stay in
 * [0.9,1.1]; mut/muu/muv ~ 9e4 +- 1 %; t,t_1 ~ 300 +- 5; u,v O(10) ...
 *
 * fp32 arrays are the fp64 value rounded once to float.
 *
 * tests/test_synth.py restates this generator in numpy and checks both the
 * host and the device fill against it.
 */
#ifndef AMT_SYNTH_H
#define AMT_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define AMT_HD __host__ __device__
#else
#define AMT_HD
#endif

/* Field ids: the order of the array/scalar-array arguments of
 * SUBROUTINE advance_mu_t (module_small_step_em.f90:7-14). */
enum amt_field {
    AMT_F_WW = 0, AMT_F_WW_1, AMT_F_U, AMT_F_U_1, AMT_F_V, AMT_F_V_1,
    AMT_F_MU, AMT_F_MUT, AMT_F_MUAVE, AMT_F_MUTS, AMT_F_MUU, AMT_F_MUV,
    AMT_F_MUDF, AMT_F_T, AMT_F_T_1, AMT_F_T_AVE, AMT_F_FT, AMT_F_MU_TEND,
    AMT_F_DNW, AMT_F_FNM, AMT_F_FNP, AMT_F_RDNW,
    AMT_F_MSFUY, AMT_F_MSFVX_INV, AMT_F_MSFTX, AMT_F_MSFTY,
    AMT_F_COUNT
};

/* rank of a field: 3 = (i,k,j), 2 = (i,j), 1 = (k) */
static inline AMT_HD int amt_field_rank(int f)
{
    switch (f) {
    case AMT_F_WW: case AMT_F_WW_1: case AMT_F_U: case AMT_F_U_1: case AMT_F_V:
    case AMT_F_V_1: case AMT_F_T: case AMT_F_T_1: case AMT_F_T_AVE: case AMT_F_FT:
        return 3;
    case AMT_F_DNW: case AMT_F_FNM: case AMT_F_FNP: case AMT_F_RDNW:
        return 1;
    default:
        return 2;
    }
}

static inline AMT_HD uint64_t amt_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* triangle wave of an index with power-of-two period P, exact in double, in [0,1] */
static inline AMT_HD double amt_tri(long n, long P)
{
    long m = n & (P - 1);
    long d = 2 * m - P;
    if (d < 0) d = -d;
    return (double)d / (double)P;
}

/* uniform in [-1,1), exact in double */
static inline AMT_HD double amt_noise(int field, uint64_t seed, uint64_t lin)
{
    uint64_t h = amt_splitmix64(seed * 0x9E3779B97F4A7C15ull
                                + (uint64_t)(field + 1) * 0xD1B54A32D192ED03ull + lin);
    double r = (double)(h >> 11) * (1.0 / 9007199254740992.0); /* 2^-53 */
    return 2.0 * r - 1.0;
}

/*
 * Value of `field` at GLOBAL zero-based memory index (gi,gk,gj) of a domain whose
 * GLOBAL memory extents are gidim x gkdim x gjdim (i.e. gi = i - ims_global ...).
 * For rank-2 fields gk is ignored, for rank-1 fields gi and gj are ignored.
 */
static inline AMT_HD double amt_synth_value(int field, uint64_t seed,
                                            long gi, long gk, long gj,
                                            long gidim, long gkdim, long gjdim)
{
    (void)gjdim;
    const int rank = amt_field_rank(field);
    if (rank == 1) {
        /* vertical coordinate metrics: dnw < 0, rdnw = 1/dnw, fnm + fnp = 1 */
        const long nk = gkdim > 1 ? gkdim - 1 : 1;
        const double wob = amt_tri(gk, 8) - 0.5;                /* [-0.5,0.5] */
        const double dnw = -(1.0 / (double)nk) * (1.0 + 0.25 * wob);
        const double fnm = 0.5 + 0.125 * (amt_tri(gk + 3, 16) - 0.5);
        switch (field) {
        case AMT_F_DNW:  return dnw;
        case AMT_F_RDNW: return 1.0 / dnw;
        case AMT_F_FNM:  return fnm;
        default:         return 1.0 - fnm;                      /* AMT_F_FNP */
        }
    }
    uint64_t lin;
    double smooth;
    if (rank == 3) {
        lin = ((uint64_t)gj * (uint64_t)gkdim + (uint64_t)gk) * (uint64_t)gidim + (uint64_t)gi;
        smooth = amt_tri(gi + 5 * field, 64) + amt_tri(gj + 11 * field, 32)
               + amt_tri(gk + 3 * field, 16) - 1.5;             /* [-1.5,1.5] */
    } else {
        lin = (uint64_t)gj * (uint64_t)gidim + (uint64_t)gi;
        smooth = amt_tri(gi + 7 * field, 128) + amt_tri(gj + 13 * field, 64) - 1.0; /* [-1,1] */
    }
    const double s = amt_noise(field, seed, lin);
    double base, asm_, ano;   /* value = base + asm_*smooth + ano*s */
    switch (field) {
    case AMT_F_U:         base = 10.0;   asm_ = 4.0;    ano = 0.1;    break;
    case AMT_F_V:         base = -6.0;   asm_ = 3.0;    ano = 0.1;    break;
    case AMT_F_U_1:       base = 1.0e-4; asm_ = 4.0e-5; ano = 1.0e-5; break;
    case AMT_F_V_1:       base = -7.0e-5; asm_ = 3.0e-5; ano = 1.0e-5; break;
    case AMT_F_T:         base = 300.0;  asm_ = 3.0;    ano = 0.5;    break;
    case AMT_F_T_1:       base = 299.0;  asm_ = 3.0;    ano = 0.5;    break;
    case AMT_F_T_AVE:     base = -777.0; asm_ = 0.0;    ano = 1.0;    break; /* sentinel: overwritten */
    case AMT_F_FT:        base = 0.0;    asm_ = 3.0e-3; ano = 1.0e-2; break;
    case AMT_F_WW:        base = 0.0;    asm_ = 0.02;   ano = 0.002;  break;
    case AMT_F_WW_1:      base = 0.0;    asm_ = 0.015;  ano = 0.002;  break;
    case AMT_F_MU:        base = 10.0;   asm_ = 3.0;    ano = 2.0;    break;
    case AMT_F_MUT:       base = 9.0e4;  asm_ = 600.0;  ano = 300.0;  break;
    case AMT_F_MUU:       base = 9.0e4;  asm_ = 600.0;  ano = 300.0;  break;
    case AMT_F_MUV:       base = 9.0e4;  asm_ = 600.0;  ano = 300.0;  break;
    case AMT_F_MU_TEND:   base = 0.0;    asm_ = 3.0e-3; ano = 1.0e-2; break;
    case AMT_F_MUAVE:     base = -555.0; asm_ = 0.0;    ano = 1.0;    break; /* sentinels: INTENT(OUT) */
    case AMT_F_MUTS:      base = -444.0; asm_ = 0.0;    ano = 1.0;    break;
    case AMT_F_MUDF:      base = -333.0; asm_ = 0.0;    ano = 1.0;    break;
    case AMT_F_MSFUY:     base = 1.0;    asm_ = 0.06;   ano = 0.04;   break;
    case AMT_F_MSFVX_INV: base = 1.0;    asm_ = 0.06;   ano = 0.04;   break;
    case AMT_F_MSFTX:     base = 1.0;    asm_ = 0.06;   ano = 0.04;   break;
    default:              base = 1.0;    asm_ = 0.06;   ano = 0.04;   break; /* AMT_F_MSFTY */
    }
    return base + asm_ * smooth + ano * s;
}

/* the four real scalars of the call */
#define AMT_SYNTH_RDX   1.0e-3
#define AMT_SYNTH_RDY   1.25e-3
#define AMT_SYNTH_DTS   2.0
#define AMT_SYNTH_EPSSM 0.1

#endif /* AMT_SYNTH_H */
