// advance_mu_t_replay -- replay a WRF dump directory through the MI355X advance_mu_t.
//
// Reads the on-disk format of the reference's drivers -- one raw stream file per variable, every
// element a BIG-ENDIAN 4-byte int or IEEE float, arrays i-fastest over the full memory extent
// (advance_mu_t_driver.f90:330,364; byte swap in common.cu:236-245; file names and order:
// advance_mu_t_driver.c:60-219) -- calls the one-shot fp32 drop-in, and prints the reference's
// per-array comparison report against the golden "*_output.bin" files (metrics of
// common.cu:68-164 / advance_mu_t_driver.f90:288-300: equal / non-equal counts, max relative and
// absolute error, max ULP distance, RMSE; the INTENT(OUT) arrays muave, muts, mudf only over the
// compute window, advance_mu_t_driver.f90:219-231).  The reference's own data set
// (/data2/WRFV3_Input_Output/V3.4.1/dyn_em/advance_mu_t/) is not shipped; the tests generate
// directories in this format from the synthetic inputs.
//
//   advance_mu_t_replay INPUT_DIR [GOLDEN_DIR] [--write OUT_DIR]
//
// Exit status: 0 ok (and, with GOLDEN_DIR, every array bit-equal), 1 usage / I/O / library error,
// 3 some array differs from its golden file.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/time.h>

#include "../../include/amt_advance_mu_t.h"

static uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

static bool read_be32(const std::string &path, void *dst, size_t count)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); return false; }
    uint32_t *d = static_cast<uint32_t *>(dst);
    const size_t got = fread(d, 4, count, f);
    fclose(f);
    if (got != count) { fprintf(stderr, "%s: expected %zu elements, found %zu\n", path.c_str(), count, got); return false; }
    for (size_t n = 0; n < count; ++n) d[n] = bswap32(d[n]);
    return true;
}

static bool write_be32(const std::string &path, const void *src, size_t count)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot create %s\n", path.c_str()); return false; }
    const uint32_t *s = static_cast<const uint32_t *>(src);
    std::vector<uint32_t> buf(1 << 16);
    for (size_t n = 0; n < count;) {
        const size_t m = std::min(buf.size(), count - n);
        for (size_t q = 0; q < m; ++q) buf[q] = bswap32(s[n + q]);
        if (fwrite(buf.data(), 4, m, f) != m) { fclose(f); return false; }
        n += m;
    }
    fclose(f);
    return true;
}

static bool exists(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (f) fclose(f);
    return f != nullptr;
}

// ULP distance of two floats: sign-magnitude bit patterns mapped onto a monotone integer line
static int64_t ulp_distance(float a, float b)
{
    int32_t ia, ib;
    memcpy(&ia, &a, 4);
    memcpy(&ib, &b, 4);
    const int64_t la = ia < 0 ? (int64_t)INT32_MIN - ia : ia;
    const int64_t lb = ib < 0 ? (int64_t)INT32_MIN - ib : ib;
    return la > lb ? la - lb : lb - la;
}

struct Window { int i0, i1, k0, k1, j0, j1; };   // zero-based memory indices, inclusive

// returns the number of non-equal values inside the window, -1 on a NaN
static long compare_report(const char *name, const float *got, const float *want,
                           int idim, int kdim, int jdim, const Window &w)
{
    long equal = 0, differ = 0;
    double max_rel = 0, max_abs = 0, sq = 0;
    int64_t max_ulp = 0;
    for (int j = w.j0; j <= w.j1; ++j)
        for (int k = w.k0; k <= w.k1; ++k)
            for (int i = w.i0; i <= w.i1; ++i) {
                const size_t e = ((size_t)j * kdim + k) * idim + i;
                const float a = got[e], b = want[e];
                if (std::isnan(a) || std::isnan(b)) {
                    printf("compare '%s': NaN at Fortran-order element (%d,%d,%d)\n", name, i, k, j);
                    return -1;
                }
                const double fa = std::fabs((double)a), fb = std::fabs((double)b);
                const double abs_err = std::fabs((double)a - (double)b);
                const double rel = (fa != 0 && fb != 0) ? abs_err / std::max(fa, fb) : std::max(fa, fb);
                max_rel = std::max(max_rel, rel);
                max_abs = std::max(max_abs, abs_err);
                max_ulp = std::max(max_ulp, ulp_distance(a, b));
                sq += abs_err * abs_err;
                if (a == b) ++equal; else ++differ;
            }
    (void)jdim;
    const double rmse = (equal + differ) ? std::sqrt(sq / (double)(equal + differ)) : 0.0;
    printf("\n# of equal values: %ld, # of non-equal values: %ld\n", equal, differ);
    printf("max relative error: %e\tmax absolute error: %e\t%s\n", max_rel, max_abs, name);
    printf("max ulp = %lld\t\t\t\trmse = %e\n", (long long)max_ulp, rmse);
    return differ;
}

int main(int argc, char **argv)
{
    std::string in, gold, out;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "--write") && a + 1 < argc) out = argv[++a];
        else if (in.empty()) in = argv[a];
        else if (gold.empty()) gold = argv[a];
        else { in.clear(); break; }
    }
    if (in.empty()) {
        fprintf(stderr, "usage: %s INPUT_DIR [GOLDEN_DIR] [--write OUT_DIR]\n", argv[0]);
        return 1;
    }
    auto path = [](const std::string &dir, const char *name) { return dir + "/" + name; };

    // ---- dimensions, scalars, flags (advance_mu_t_driver.c:60-137) ----
    const char *dim_names[17] = {"ids", "ide", "jds", "jde", "kde", "ims", "ime", "jms", "jme", "kms", "kme",
                                 "its", "ite", "jts", "jte", "kts", "kte"};
    int dims[17];
    for (int n = 0; n < 17; ++n)
        if (!read_be32(path(in, (std::string(dim_names[n]) + ".bin").c_str()), &dims[n], 1)) return 1;
    const int ids = dims[0], ide = dims[1], jds = dims[2], jde = dims[3], kde = dims[4], ims = dims[5],
              ime = dims[6], jms = dims[7], jme = dims[8], kms = dims[9], kme = dims[10], its = dims[11],
              ite = dims[12], jts = dims[13], jte = dims[14], kts = dims[15], kte = dims[16];
    if (exists(path(in, "kds.bin"))) {               // only the C/CUDA drivers have it (advance_mu_t_driver.c:64)
        int kds = 0;
        if (!read_be32(path(in, "kds.bin"), &kds, 1)) return 1;
        if (kds != kts) fprintf(stderr, "note: kds=%d differs from kts=%d; the Fortran routine has no kds\n", kds, kts);
    }
    float rdx, rdy, dts, epssm;
    if (!read_be32(path(in, "grid_rdx.bin"), &rdx, 1) || !read_be32(path(in, "grid_rdy.bin"), &rdy, 1) ||
        !read_be32(path(in, "dts_rk.bin"), &dts, 1) || !read_be32(path(in, "grid_epssm.bin"), &epssm, 1)) return 1;
    int nested, periodic_x, specified;
    if (!read_be32(path(in, "config_flags_nested.bin"), &nested, 1) ||
        !read_be32(path(in, "config_flags_periodic_x.bin"), &periodic_x, 1) ||
        !read_be32(path(in, "config_flags_specified.bin"), &specified, 1)) return 1;

    const int idim = ime - ims + 1, kdim = kme - kms + 1, jdim = jme - jms + 1;
    if (idim < 1 || kdim < 1 || jdim < 1) { fprintf(stderr, "bad memory extents\n"); return 1; }
    const size_t n3 = (size_t)idim * kdim * jdim, n2 = (size_t)idim * jdim, n1 = (size_t)kdim;

    // ---- arrays (file names: advance_mu_t_driver.c:149-219) ----
    std::vector<float> dnw(n1), fnm(n1), fnp(n1), rdnw(n1);
    std::vector<float> mut(n2), muu(n2), muv(n2), mu_tend(n2), msfuy(n2), msfvx_inv(n2), msfty(n2), msftx(n2), mu(n2);
    std::vector<float> muave(n2, 0.f), muts(n2, 0.f), mudf(n2, 0.f);           // INTENT(OUT): not read
    std::vector<float> u(n3), u_1(n3), v(n3), v_1(n3), t_1(n3), ft(n3), ww(n3), ww_1(n3), t(n3), t_ave(n3);
    struct In { const char *file; std::vector<float> *a; };
    const In ins[] = {
        {"grid_dnw.bin", &dnw}, {"grid_fnm.bin", &fnm}, {"grid_fnp.bin", &fnp}, {"grid_rdnw.bin", &rdnw},
        {"grid_mut.bin", &mut}, {"grid_muu.bin", &muu}, {"grid_muv.bin", &muv}, {"mu_tend.bin", &mu_tend},
        {"grid_msfuy.bin", &msfuy}, {"grid_msfvx_inv.bin", &msfvx_inv}, {"grid_msfty.bin", &msfty},
        {"grid_msftx.bin", &msftx}, {"grid_mu_2.bin", &mu},
        {"grid_u_2.bin", &u}, {"grid_u_save.bin", &u_1}, {"grid_v_2.bin", &v}, {"grid_v_save.bin", &v_1},
        {"grid_t_save.bin", &t_1}, {"t_tend.bin", &ft},
        {"grid_ww.bin", &ww}, {"ww1.bin", &ww_1}, {"grid_t_2.bin", &t}, {"t_2save.bin", &t_ave},
    };
    for (const In &x : ins)
        if (!read_be32(path(in, x.file), x.a->data(), x.a->size())) return 1;
    printf("advance_mu_t replay: memory %dx%dx%d (i,k,j), tile i %d:%d j %d:%d k %d:%d, flags nested=%d periodic_x=%d specified=%d\n",
           idim, kdim, jdim, its, ite, jts, jte, kts, kte, nested, periodic_x, specified);

    // ---- the call the reference drivers time (advance_mu_t_driver.c:222-245) ----
    timeval t0, t1;
    gettimeofday(&t0, nullptr);
    const int rc = amt_advance_mu_t_f32(ww.data(), ww_1.data(), u.data(), u_1.data(), v.data(), v_1.data(),
                                        mu.data(), mut.data(), muave.data(), muts.data(), muu.data(), muv.data(),
                                        mudf.data(), t.data(), t_1.data(), t_ave.data(), ft.data(), mu_tend.data(),
                                        rdx, rdy, dts, epssm, dnw.data(), fnm.data(), fnp.data(), rdnw.data(),
                                        msfuy.data(), msfvx_inv.data(), msftx.data(), msfty.data(),
                                        periodic_x, specified, nested, ids, ide, jds, jde, kde,
                                        ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte);
    gettimeofday(&t1, nullptr);
    if (rc != AMT_OK) { fprintf(stderr, "advance_mu_t failed (status %d): %s\n", rc, amt_last_error()); return 1; }
    printf("advance_mu_t computing time(msec): %f\n",
           (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) * 1e-3);

    struct Out { const char *file; const std::vector<float> *a; int rank; bool window_only; };
    const Out outs[] = {
        {"grid_ww_output.bin", &ww, 3, false}, {"ww1_output.bin", &ww_1, 3, false},
        {"grid_t_2_output.bin", &t, 3, false}, {"t_2save_output.bin", &t_ave, 3, false},
        {"grid_mu_2_output.bin", &mu, 2, false}, {"muave_output.bin", &muave, 2, true},
        {"grid_muts_output.bin", &muts, 2, true}, {"grid_mudf_output.bin", &mudf, 2, true},
    };
    if (!out.empty())
        for (const Out &o : outs)
            if (!write_be32(path(out, o.file), o.a->data(), o.a->size())) return 1;

    if (gold.empty()) return 0;
    int i_start, i_end, j_start, j_end, k_start, k_end;
    amt_compute_window(periodic_x, specified, nested, ids, ide, jds, jde, its, ite, jts, jte, kts, kte,
                       &i_start, &i_end, &j_start, &j_end, &k_start, &k_end);
    long bad = 0;
    for (const Out &o : outs) {
        std::vector<float> want(o.a->size());
        if (!read_be32(path(gold, o.file), want.data(), want.size())) return 1;
        Window w{0, idim - 1, 0, (o.rank == 3 ? kdim : 1) - 1, 0, jdim - 1};
        if (o.window_only) { w.i0 = i_start - ims; w.i1 = i_end - ims; w.j0 = j_start - jms; w.j1 = j_end - jms; }
        const long d = compare_report(o.file, o.a->data(), want.data(), idim, o.rank == 3 ? kdim : 1, jdim, w);
        if (d != 0) bad += (d < 0 ? 1 : d);
    }
    printf("\n%s\n", bad ? "RESULT: differs from the golden files" : "RESULT: all 8 arrays bit-equal to the golden files");
    return bad ? 3 : 0;
}
