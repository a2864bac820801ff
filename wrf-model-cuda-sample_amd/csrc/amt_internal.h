// amt_internal.h -- what the translation units of the host-side runtime share (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <new>

#include "../../include/amt_advance_mu_t.h"
#include "../../include/amt_synth.h"
#include "amt_params.h"

// ---------------------------------------------------------------------------
// errors: the text goes to the calling thread's amt_last_error(), the status is returned
// ---------------------------------------------------------------------------
int amt_fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define AMT_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return amt_fail(e_ == hipErrorNoDevice ? AMT_ERR_NO_DEVICE : AMT_ERR_HIP,   \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                            __FILE__, __LINE__);                                        \
    } while (0)

// ---------------------------------------------------------------------------
// Fault injection for the halo-freshness tests (tests/test_gpu_34_halo_freshness.py): AMT_TEST_FAULT="<step>@<n>" turns the
// n-th occurrence (1-based, per exchange / per stepper) of one step of the halo exchange into a no-op, everything around it
// -- sequence numbers, posts, waits -- running as usual, so that the sweep completes with STALE halo data instead of
// hanging.  Steps: skip_stage (IPC: the staging copies of the send segments are not refreshed), skip_pull (IPC: the rows
// are not pulled), skip_group (RCCL: the ncclSend/ncclRecv group is not issued -- every rank must carry the same setting),
// skip_pack / skip_unpack (grid stepper: the halo columns are not gathered / scattered).  A test that cannot tell such a
// run from a healthy one does not test the exchange beyond its first sweep.  Unset (always, outside those tests): no effect.
// ---------------------------------------------------------------------------
inline bool amt_test_fault(const char *step, unsigned long long occurrence)
{
    const char *spec = getenv("AMT_TEST_FAULT");          // read per call (a few times per sweep): a test sets and clears it in-process
    if (!spec || !*spec) return false;
    const char *at = strchr(spec, '@');
    if (!at || (size_t)(at - spec) != strlen(step) || strncmp(spec, step, at - spec) != 0) return false;
    return strtoull(at + 1, nullptr, 10) == occurrence;
}

// ---------------------------------------------------------------------------
// argument bundle shared by the entry points
// ---------------------------------------------------------------------------
template <typename T>
struct AmtArgs {
    T *ww; const T *ww_1, *u, *u_1, *v, *v_1;
    T *mu; const T *mut; T *muave, *muts; const T *muu, *muv;
    T *mudf, *t; const T *t_1; T *t_ave; const T *ft, *mu_tend;
    T rdx, rdy, dts, epssm;
    const T *dnw, *fnm, *fnp, *rdnw, *msfuy, *msfvx_inv, *msftx, *msfty;
    int periodic_x, specified, nested;
    int ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte;
};


// Checks the preconditions and rebases the Fortran bounds to memory-relative zero-based ones;
// *empty is set when the compute window holds no column (amt_api.hip).
template <typename T> int amt_build_params(const AmtArgs<T> &a, AmtParams<T> &p, AmtWindow &w, bool *empty);
// bounds check + kernel launch on a stream: the device-resident entry point (amt_api.hip)
template <typename T> int amt_device_call(void *hip_stream, int variant, const AmtArgs<T> &a);
// the same for the first and the last row of the window only (one launch where the march kernel runs)
template <typename T> int amt_device_call_edges(void *hip_stream, int variant, const AmtArgs<T> &a);
extern template int amt_build_params<float>(const AmtArgs<float> &, AmtParams<float> &, AmtWindow &, bool *);
extern template int amt_build_params<double>(const AmtArgs<double> &, AmtParams<double> &, AmtWindow &, bool *);
extern template int amt_device_call<float>(void *, int, const AmtArgs<float> &);
extern template int amt_device_call<double>(void *, int, const AmtArgs<double> &);
// the whole window, launched beside another stream's kernels (planned in at least two rounds of workgroups)
template <typename T> int amt_device_call_shared(void *hip_stream, int variant, const AmtArgs<T> &a);
extern template int amt_device_call_shared<float>(void *, int, const AmtArgs<float> &);
extern template int amt_device_call_shared<double>(void *, int, const AmtArgs<double> &);
extern template int amt_device_call_edges<float>(void *, int, const AmtArgs<float> &);
extern template int amt_device_call_edges<double>(void *, int, const AmtArgs<double> &);

#define AMT_PACK_ARGS(T)                                                                        \
    AmtArgs<T> a;                                                                               \
    a.ww = ww; a.ww_1 = ww_1; a.u = u; a.u_1 = u_1; a.v = v; a.v_1 = v_1; a.mu = mu;            \
    a.mut = mut; a.muave = muave; a.muts = muts; a.muu = muu; a.muv = muv; a.mudf = mudf;       \
    a.t = t; a.t_1 = t_1; a.t_ave = t_ave; a.ft = ft; a.mu_tend = mu_tend;                      \
    a.rdx = rdx; a.rdy = rdy; a.dts = dts; a.epssm = epssm;                                     \
    a.dnw = dnw; a.fnm = fnm; a.fnp = fnp; a.rdnw = rdnw; a.msfuy = msfuy;                      \
    a.msfvx_inv = msfvx_inv; a.msftx = msftx; a.msfty = msfty;                                  \
    a.periodic_x = periodic_x; a.specified = specified; a.nested = nested;                      \
    a.ids = ids; a.ide = ide; a.jds = jds; a.jde = jde; a.kde = kde;                            \
    a.ims = ims; a.ime = ime; a.jms = jms; a.jme = jme; a.kms = kms; a.kme = kme;               \
    a.its = its; a.ite = ite; a.jts = jts; a.jte = jte; a.kts = kts; a.kte = kte;

#define AMT_SIG(T)                                                                              \
    T *ww, const T *ww_1, const T *u, const T *u_1, const T *v, const T *v_1,                   \
    T *mu, const T *mut, T *muave, T *muts, const T *muu, const T *muv,                         \
    T *mudf, T *t, const T *t_1, T *t_ave, const T *ft, const T *mu_tend,                       \
    T rdx, T rdy, T dts, T epssm,                                                               \
    const T *dnw, const T *fnm, const T *fnp, const T *rdnw,                                    \
    const T *msfuy, const T *msfvx_inv, const T *msftx, const T *msfty,                         \
    int periodic_x, int specified, int nested,                                                  \
    int ids, int ide, int jds, int jde, int kde,                                                \
    int ims, int ime, int jms, int jme, int kms, int kme,                                       \
    int its, int ite, int jts, int jte, int kts, int kte


// ---------------------------------------------------------------------------
// resident domain handle (amt_domain.hip), also stepped by the slab stepper (amt_slab.hip)
// ---------------------------------------------------------------------------
namespace {
// makes the domain's device current for the duration of a call and restores the caller's
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int want)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != want) switched = (hipSetDevice(want) == hipSuccess);
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};
}  // namespace

struct amt_domain {
    int dtype_bytes = 8;
    int periodic_x = 0, specified = 0, nested = 0;
    int ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte;
    double rdx = AMT_SYNTH_RDX, rdy = AMT_SYNTH_RDY, dts = AMT_SYNTH_DTS, epssm = AMT_SYNTH_EPSSM;
    int variant = AMT_VARIANT_AUTO;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *field[AMT_F_COUNT] = {};
    bool owns_fields = true;      // false: the arrays belong to the caller (amt_domain_wrap)
    bool owns_stream = true;      // false: the stream belongs to the caller
    int placement_tries = 0;      // allocations of the state that were timed (amt_domain_create / amt_domain_tune_placement)
    float placement_ms[16] = {};  // sweep time on each (0: not tried)
    size_t count(int f) const
    {
        const size_t idim = ime - ims + 1, kdim = kme - kms + 1, jdim = jme - jms + 1;
        const int r = amt_field_rank(f);
        return r == 3 ? idim * kdim * jdim : r == 2 ? idim * jdim : kdim;
    }
};

template <typename T>
inline void amt_domain_args(amt_domain *d, AmtArgs<T> &a)
{
    T **f = reinterpret_cast<T **>(d->field);
    a.ww = f[AMT_F_WW]; a.ww_1 = f[AMT_F_WW_1]; a.u = f[AMT_F_U]; a.u_1 = f[AMT_F_U_1];
    a.v = f[AMT_F_V]; a.v_1 = f[AMT_F_V_1]; a.mu = f[AMT_F_MU]; a.mut = f[AMT_F_MUT];
    a.muave = f[AMT_F_MUAVE]; a.muts = f[AMT_F_MUTS]; a.muu = f[AMT_F_MUU]; a.muv = f[AMT_F_MUV];
    a.mudf = f[AMT_F_MUDF]; a.t = f[AMT_F_T]; a.t_1 = f[AMT_F_T_1]; a.t_ave = f[AMT_F_T_AVE];
    a.ft = f[AMT_F_FT]; a.mu_tend = f[AMT_F_MU_TEND];
    a.rdx = (T)d->rdx; a.rdy = (T)d->rdy; a.dts = (T)d->dts; a.epssm = (T)d->epssm;
    a.dnw = f[AMT_F_DNW]; a.fnm = f[AMT_F_FNM]; a.fnp = f[AMT_F_FNP]; a.rdnw = f[AMT_F_RDNW];
    a.msfuy = f[AMT_F_MSFUY]; a.msfvx_inv = f[AMT_F_MSFVX_INV]; a.msftx = f[AMT_F_MSFTX];
    a.msfty = f[AMT_F_MSFTY];
    a.periodic_x = d->periodic_x; a.specified = d->specified; a.nested = d->nested;
    a.ids = d->ids; a.ide = d->ide; a.jds = d->jds; a.jde = d->jde; a.kde = d->kde;
    a.ims = d->ims; a.ime = d->ime; a.jms = d->jms; a.jme = d->jme; a.kms = d->kms; a.kme = d->kme;
    a.its = d->its; a.ite = d->ite; a.jts = d->jts; a.jte = d->jte; a.kts = d->kts; a.kte = d->kte;
}
