/*
 * include/amt_advance_mu_t.h -- C-ABI of the MI355X (gfx950) advance_mu_t path.
 *
 * Drop-in boundary for WRF-ARW's acoustic-substep routine
 *   SUBROUTINE advance_mu_t(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu,
 *                           muv, mudf, t, t_1, t_ave, ft, mu_tend, rdx, rdy, dts,
 *                           epssm, dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx,
 *                           msfty, config_flags, ids,ide, jds,jde, kde, ims,ime,
 *                           jms,jme, kms,kme, its,ite, jts,jte, kts,kte)
 * (reference: module_small_step_em.f90:7-18; C/CUDA precedent for a flat C
 * signature: advance_mu_t.h:10-23, advance_mu_t_cu.h:3-17).
 *
 * Conventions shared by every entry point
 *  - Argument ORDER is the Fortran one.  config_flags is replaced by the three
 *    logicals the routine reads (module_small_step_em.f90:97-103; the C struct
 *    of advance_mu_t.h:3-8 carries the same three) as int 0/1, in the order
 *    periodic_x, specified, nested.
 *  - All 17 bounds are Fortran-style INCLUSIVE indices passed unchanged; there
 *    is no kds (the Fortran signature has none).  Arrays are i-fastest:
 *    element (i,k,j) of a 3-D array is at ((j-jms)*kdim + (k-kms))*idim + (i-ims),
 *    (i,j) of a 2-D array at (j-jms)*idim + (i-ims), (k) of a 1-D array at k-kms,
 *    with idim = ime-ims+1, kdim = kme-kms+1  (advance_mu_t.c:8-9,33-55).
 *  - Preconditions (anything else is undefined in the Fortran as well, see
 *    DESIGN.md): kts == 1, kte == kde, kms <= 1, kme >= kte, and the compute
 *    window plus its one-cell input halo (i-1, i+1, j-1, j+1) inside memory.
 *    Level count: a column's k chains live in LDS, which holds 320 levels in
 *    fp64 and 640 in fp32 (AMT_ERR_PRECONDITION beyond; WRF runs 30..150).
 *    SPEED CLIFF (a step since r06): the production kernel (AMT_VARIANT_MARCH) keeps four level-by-column
 *    buffers of a tile in the CU's 160 KB of LDS, which ends at 240 levels in fp64 and
 *    264 in fp32 (16-column tiles; a fifth level per lane does not fit).  Beyond that
 *    AMT_VARIANT_AUTO falls back to the column kernel, which re-reads its neighbours and evaluates dvdxi twice instead of
 *    keeping it (nothing in LDS): about 0.41 of the HBM roofline against 0.61-0.75 for the march kernel (until r06: 0.09, one
 *    wave per compute unit under a 150 KB LDS column); the first such call of a process says so
 *    on stderr (AMT_QUIET=1 suppresses it).
 *  - Return value: AMT_OK or an amt_status code; nothing ever calls exit()
 *    (the reference's wrapper prints and exit(1)s, advance_mu_t_no_async.cu:22-32,
 *    82-85).  amt_last_error() gives the text for the calling thread.
 *  - Outputs outside the compute window, and level k = kte, keep the caller's
 *    contents (the reference uploads its OUT arrays for the same reason,
 *    advance_mu_t_no_async.cu:259,270-272).
 *  - Calls on the same stream/handle are not re-entrant; different streams are
 *    independent.
 *
 * There is no CPU fallback anywhere behind this header: without a HIP device
 * every compute entry point returns AMT_ERR_NO_DEVICE / AMT_ERR_HIP.
 */
#ifndef AMT_ADVANCE_MU_T_H
#define AMT_ADVANCE_MU_T_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum amt_status {
    AMT_OK = 0,
    AMT_ERR_HIP = 1,           /* a HIP runtime call failed (text in amt_last_error) */
    AMT_ERR_PRECONDITION = 2,  /* bounds violate the preconditions above            */
    AMT_ERR_INVALID_ARG = 3,   /* null pointer, bad dtype / field id / variant      */
    AMT_ERR_NO_DEVICE = 4,     /* no gfx950 device visible                          */
    AMT_ERR_ALLOC = 5,         /* device or host allocation failed                  */
    AMT_ERR_COMM = 6           /* RCCL could not be loaded or a collective call failed */
} amt_status;

/* Kernel variants (amt_set_variant / `variant` arguments).  AMT_VARIANT_AUTO picks
 * the fastest one that supports the given shape. */
enum amt_variant {
    AMT_VARIANT_AUTO = 0,
    AMT_VARIANT_COLUMN = 1,    /* one lane per (i,j) column, k-column staged in LDS */
    AMT_VARIANT_MARCH = 2      /* (i,k)-cell lanes marching in j, k-chains through LDS */
};
/* OR-ed into the `variant` argument of amt_advance_mu_t_device_*: kernels of ANOTHER stream are meant to run beside this
 * launch (a j-slab's interior rows while its halo exchange and edge rows go through a communication stream).  On its own a
 * launch may be planned as ONE round of workgroups that holds every compute unit until it ends; with this flag it is
 * planned in at least two rounds, so that the other stream's kernels get compute units at a round boundary.  Same bits. */
#define AMT_LAUNCH_BESIDE_OTHERS 0x100

const char *amt_version(void);
const char *amt_status_string(int status);
const char *amt_last_error(void);
int amt_device_count(void);               /* number of visible HIP devices, 0 if none */

/* ------------------------------------------------------------------------
 * (1) One-shot drop-ins: HOST arrays in, HOST arrays out.  Replaces the body of
 *     the Fortran routine (module_small_step_em.f90:7-252) and the CUDA host
 *     wrapper (advance_mu_t_no_async.cu:35-424): allocate, upload, launch,
 *     download, free -- on the current HIP device.  Of ww, t and t_ave only the
 *     cells the Fortran assigns (i_start..i_end, 1..kte-1, j_start..j_end) are
 *     written back; t_ave and the levels above 1 of ww are outputs only and are
 *     not uploaded.  Of the 2-D outputs, too, only the window's cells are written
 *     back: host threads that run tiles of one domain concurrently (OpenMP tiles,
 *     split in i or in j) never touch each other's cells.
 * ------------------------------------------------------------------------ */
int amt_advance_mu_t_f32(
    float *ww, const float *ww_1, const float *u, const float *u_1,
    const float *v, const float *v_1,
    float *mu, const float *mut, float *muave, float *muts,
    const float *muu, const float *muv,
    float *mudf, float *t, const float *t_1,
    float *t_ave, const float *ft, const float *mu_tend,
    float rdx, float rdy, float dts, float epssm,
    const float *dnw, const float *fnm, const float *fnp, const float *rdnw,
    const float *msfuy, const float *msfvx_inv,
    const float *msftx, const float *msfty,
    int periodic_x, int specified, int nested,
    int ids, int ide, int jds, int jde, int kde,
    int ims, int ime, int jms, int jme, int kms, int kme,
    int its, int ite, int jts, int jte, int kts, int kte);

int amt_advance_mu_t_f64(
    double *ww, const double *ww_1, const double *u, const double *u_1,
    const double *v, const double *v_1,
    double *mu, const double *mut, double *muave, double *muts,
    const double *muu, const double *muv,
    double *mudf, double *t, const double *t_1,
    double *t_ave, const double *ft, const double *mu_tend,
    double rdx, double rdy, double dts, double epssm,
    const double *dnw, const double *fnm, const double *fnp, const double *rdnw,
    const double *msfuy, const double *msfvx_inv,
    const double *msftx, const double *msfty,
    int periodic_x, int specified, int nested,
    int ids, int ide, int jds, int jde, int kde,
    int ims, int ime, int jms, int jme, int kms, int kme,
    int its, int ite, int jts, int jte, int kts, int kte);

/* Page-lock / release a host array (hipHostRegister).  The one-shot calls stream the window in
 * j chunks through an upload, a compute and a download stream so that both directions of the
 * host link and the kernels overlap.  With page-locked 3-D arrays (by these, or allocated pinned
 * by the caller as the reference driver does, advance_mu_t_driver.cu:97-167) every copy is
 * asynchronous; with pageable ones a second host thread issues the downloads and small arrays
 * pass through an internal page-locked staging buffer -- about the same speed for large
 * domains, ~2x slower than pinned for patch-sized ones.  Pin once, outside the time loop. */
int amt_host_pin(void *ptr, size_t bytes);
int amt_host_unpin(void *ptr);

/* One call, several devices.  The reference's host call IS its multi-GPU call: advance_mu_t_no_async.cu:108-162 splits j over
 * its `GPUs` devices inside one advance_mu_t(...), refills every device's halo rows from the host arrays (:135-160), launches and
 * gathers per device (:329-390).  amt_host_set_devices(n, ids) gives the calling host thread n DEVICE SLOTS (ids may name a device
 * more than once); from then on each one-shot call of that thread cuts its tile's rows jts..jte into n contiguous pieces (uneven
 * where they do not divide) and runs them at the same time, one per slot, each piece a tile of the same domain whose halo rows
 * come from the HOST arrays exactly as the reference's do -- no traffic between the devices, same bits as the one-device call.
 * The one-shot path is bound by the host link; a node has one link per device.  Every slot keeps its
 * own workspace, residency cache and deferred outputs on its device; the control calls below (amt_host_cache_*, _invalidate,
 * _defer, _fetch, _stale, _release) act on all slots.  n = 0 turns it off (what the slots' devices alone hold comes down first).
 * AMT_ONESHOT_DEVICES="0,1,2,3" or "all" in the environment does the same for every thread that never calls this (the Fortran
 * drop-in module is one CALL and has nowhere to put another).  amt_host_devices returns the number of slots of the calling thread
 * and their device ids. */
int amt_host_set_devices(int n, const int *device_ids);
int amt_host_devices(int *device_ids, int cap);

/* The one-shot calls keep their device workspace (three streams, six events, the buffer arena
 * when it is at most 1 GiB) per calling host thread between calls, instead of the reference
 * wrapper's allocate-and-free on every call (advance_mu_t_no_async.cu:178-244,392-423), which
 * at WRF patch sizes costs more than the call itself.  This frees the calling thread's
 * workspace now; it is freed anyway when the thread ends. */
int amt_host_release(void);

/* Residency cache of the one-shot calls (per calling host thread; off by default).  In WRF's acoustic loop
 * (solve_em) advance_mu_t is called number_of_small_timesteps times per Runge-Kutta stage with the SAME
 * ww_1, u_1, v_1, t_1 (the `_save` linearisation state) and ft (the tendency): five of the eight 3-D arrays a
 * call uploads.  With the cache enabled the calling thread keeps whole-window device copies of these five
 * between calls -- keyed on their host addresses and every extent -- and uploads one again only after
 * amt_host_invalidate(ptr) (NULL: all of them; call it when a new RK stage has rewritten them, and when an
 * array was freed and another one allocated at the same address: the key is the address, not the contents).  Since r04
 * the 2-D and 1-D inputs that are constant over the sub-steps as well (mut, muu, muv, mu_tend, the four map factors, dnw,
 * fnm, fnp, rdnw) are kept the same way.  u, v, t, mu and level 1 of ww, which change from sub-step to sub-step, and all
 * outputs cross the link on every call as before (unless deferred, below).  The reference re-uploads everything on every call
 * (advance_mu_t_no_async.cu:245-306).
 * amt_host_cache_check(1) is the debug mode: every call checksums the cached arrays on the host and fails with
 * AMT_ERR_PRECONDITION if one changed without an invalidate (it reads the whole arrays: slow).
 * amt_host_cache_enable(0) and amt_host_release() free the copies. */
int amt_host_cache_enable(int on);
int amt_host_cache_check(int on);
int amt_host_invalidate(const void *host_ptr);

/* Deferred outputs of the one-shot calls (per calling host thread; off by default): the practical form of a resident
 * small-step loop at the reference's own boundary (every call of advance_mu_t_no_async.cu brings all outputs down,
 * :366-390, and takes them up again with the next call, :245-306).  amt_host_defer(ptr, 1) marks one of the routine's
 * outputs -- ww, t, t_ave, mu, muave, muts, mudf, recognised by its host address in the calls that follow; NULL: all
 * seven -- as DEFERRED: it stays on the device after a call (a whole-window copy, like the cached inputs) and nothing of it
 * is written to the host array; the in/out ones (level 1 of ww, t, mu) are then also not uploaded by the next call --
 * the device copy is the truth.  amt_host_fetch(ptr) (NULL: every stale one) brings the window's cells down when the
 * caller -- the next routine of the acoustic loop, or the end of the loop -- really needs host values; amt_host_stale(ptr)
 * (NULL: any) says whether the device copy is newer than the host array.  amt_host_invalidate(ptr) says the HOST array
 * was rewritten and is the truth again (the next call uploads it).  With the residency cache on as well, a sub-step
 * uploads u, v and the 2-D / 1-D inputs only.  amt_host_defer(ptr, 0) fetches what is stale and makes the array an
 * ordinary output again; so do amt_host_release() and a call with other extents or other arrays (nothing the device
 * alone holds is ever dropped by the library; a thread that ENDS without amt_host_release() loses it).
 * In the debug mode (amt_host_cache_check(1)) the window's cells of a deferred host array are overwritten with NaN
 * canaries after every call, so that a consumer reading the stale array computes NaNs instead of silently using old
 * values, and a call that finds them changed without an invalidate fails with AMT_ERR_PRECONDITION.
 * After a call that FAILS once its kernels were enqueued, the device copies of the deferred outputs are undefined (t, mu and
 * level 1 of ww are advanced in place, chunk by chunk): amt_host_stale then returns -1 for them, amt_host_fetch and the next
 * call fail with AMT_ERR_PRECONDITION, and amt_host_invalidate(ptr) -- the host array as last fetched is the truth -- clears it. */
int amt_host_defer(const void *host_ptr, int on);
int amt_host_fetch(const void *host_ptr);
int amt_host_stale(const void *host_ptr);

/* ------------------------------------------------------------------------
 * (2) Device-resident drop-ins: the same call with every array pointer in
 *     DEVICE memory of the current device, enqueued on `hip_stream`
 *     (a hipStream_t; NULL = the default stream) and returning without
 *     synchronising.  This is the kernel-only boundary the reference times
 *     (advance_mu_t_no_async.cu:324-363) and what a resident small-step loop
 *     (solve_em) or a j-slab owner calls per sub-step; a slab owner passes the
 *     GLOBAL ids..jde and its LOCAL jms..jme / jts..jte, exactly as a WRF patch
 *     does.  `variant` is an amt_variant.
 * ------------------------------------------------------------------------ */
int amt_advance_mu_t_device_f32(
    void *hip_stream, int variant,
    float *ww, const float *ww_1, const float *u, const float *u_1,
    const float *v, const float *v_1,
    float *mu, const float *mut, float *muave, float *muts,
    const float *muu, const float *muv,
    float *mudf, float *t, const float *t_1,
    float *t_ave, const float *ft, const float *mu_tend,
    float rdx, float rdy, float dts, float epssm,
    const float *dnw, const float *fnm, const float *fnp, const float *rdnw,
    const float *msfuy, const float *msfvx_inv,
    const float *msftx, const float *msfty,
    int periodic_x, int specified, int nested,
    int ids, int ide, int jds, int jde, int kde,
    int ims, int ime, int jms, int jme, int kms, int kme,
    int its, int ite, int jts, int jte, int kts, int kte);

int amt_advance_mu_t_device_f64(
    void *hip_stream, int variant,
    double *ww, const double *ww_1, const double *u, const double *u_1,
    const double *v, const double *v_1,
    double *mu, const double *mut, double *muave, double *muts,
    const double *muu, const double *muv,
    double *mudf, double *t, const double *t_1,
    double *t_ave, const double *ft, const double *mu_tend,
    double rdx, double rdy, double dts, double epssm,
    const double *dnw, const double *fnm, const double *fnp, const double *rdnw,
    const double *msfuy, const double *msfvx_inv,
    const double *msftx, const double *msfty,
    int periodic_x, int specified, int nested,
    int ids, int ide, int jds, int jde, int kde,
    int ims, int ime, int jms, int jme, int kms, int kme,
    int its, int ite, int jts, int jte, int kts, int kte);

/* Compute window the routine will update for the given flags and bounds
 * (module_small_step_em.f90:91-106).  Pure host arithmetic, no device needed. */
int amt_compute_window(int periodic_x, int specified, int nested,
                       int ids, int ide, int jds, int jde,
                       int its, int ite, int jts, int jte, int kts, int kte,
                       int *i_start, int *i_end, int *j_start, int *j_end,
                       int *k_start, int *k_end);

/* ------------------------------------------------------------------------
 * (3) Resident domain handle: owns the 26 device arrays of one patch
 *     (memory extents ims:ime, kms:kme, jms:jme), a compute stream and the four
 *     scalars.  For C / Fortran hosts that keep ww, t, mu on the GPU across
 *     sub-steps; dtype_bytes is 4 (float) or 8 (double).  Field ids are
 *     enum amt_field of amt_synth.h (the Fortran argument order).
 * ------------------------------------------------------------------------ */
typedef struct amt_domain amt_domain;

int amt_domain_create(amt_domain **out, int dtype_bytes,
                      int periodic_x, int specified, int nested,
                      int ids, int ide, int jds, int jde, int kde,
                      int ims, int ime, int jms, int jme, int kms, int kme,
                      int its, int ite, int jts, int jte, int kts, int kte);
/* The same handle over device arrays the CALLER owns (a host model that already keeps its state
 * on the GPU, as WRF's arrays would be): fields[f] is the device pointer of field f (enum
 * amt_field, all AMT_F_COUNT of them, laid out as above); hip_stream is the hipStream_t the
 * handle's work is enqueued on, NULL = a stream of its own.  Nothing is copied; destroy frees
 * neither the arrays nor the caller's stream. */
int amt_domain_wrap(amt_domain **out, int dtype_bytes,
                    int periodic_x, int specified, int nested,
                    int ids, int ide, int jds, int jde, int kde,
                    int ims, int ime, int jms, int jme, int kms, int kme,
                    int its, int ite, int jts, int jte, int kts, int kte,
                    void *const *fields, void *hip_stream);
int amt_domain_destroy(amt_domain *d);
int amt_domain_set_scalars(amt_domain *d, double rdx, double rdy, double dts, double epssm);
int amt_domain_set_variant(amt_domain *d, int variant);
/* whole-array copies in the (ims:ime[,kms:kme][,jms:jme]) layout; synchronous */
int amt_domain_upload(amt_domain *d, int field, const void *host);
int amt_domain_download(amt_domain *d, int field, void *host);
/* rows j_lo..j_hi (Fortran indices inside jms:jme) of a rank-3 or rank-2 field; `host` holds
 * exactly those rows (contiguous in this layout); synchronous */
int amt_domain_upload_rows(amt_domain *d, int field, int j_lo, int j_hi, const void *host);
int amt_domain_download_rows(amt_domain *d, int field, int j_lo, int j_hi, void *host);
/* fill every field from amt_synth.h; (gi0,gk0,gj0) = GLOBAL zero-based index of this
 * patch's element (ims,kms,jms); (gidim,gkdim,gjdim) = GLOBAL memory extents */
int amt_domain_fill_synthetic(amt_domain *d, uint64_t seed,
                              long gi0, long gk0, long gj0,
                              long gidim, long gkdim, long gjdim);
/* The same for the fields whose bit is set in field_mask (bit f = enum amt_field f) only; asynchronous on the domain's
 * stream.  With AMT_EXCHANGED_FIELDS and a seed that changes per sub-step this stands in for WRF's advance_uv, which
 * rewrites u and v before every advance_mu_t call (the operands of module_small_step_em.f90:143-146, :241-245): the rows
 * and columns a patch sends then differ from sweep to sweep, and only an exchange that delivers EVERY sweep gives the
 * bits of the unsplit run.  (AMT_EXCHANGED_FIELDS names enum amt_field: include amt_synth.h where it is expanded.) */
#define AMT_FIELD_BIT(f) (1ull << (f))
#define AMT_EXCHANGED_FIELDS                                                                                   \
    (AMT_FIELD_BIT(AMT_F_U) | AMT_FIELD_BIT(AMT_F_U_1) | AMT_FIELD_BIT(AMT_F_V) | AMT_FIELD_BIT(AMT_F_V_1) |   \
     AMT_FIELD_BIT(AMT_F_T_1) | AMT_FIELD_BIT(AMT_F_MUU) | AMT_FIELD_BIT(AMT_F_MUV) | AMT_FIELD_BIT(AMT_F_MSFUY) | \
     AMT_FIELD_BIT(AMT_F_MSFVX_INV))
int amt_domain_fill_fields(amt_domain *d, uint64_t field_mask, uint64_t seed,
                           long gi0, long gk0, long gj0,
                           long gidim, long gkdim, long gjdim);
/* Verification aid: overwrite with NaN exactly what the stencil reads from a neighbour on the given sides -- row jte+1 of
 * v, v_1, t_1, muv, msfvx_inv (AMT_SIDE_ABOVE), row jts-1 of t_1 (AMT_SIDE_BELOW), column ite+1 of u, u_1, t_1, muu, msfuy
 * (AMT_SIDE_RIGHT), column its-1 of t_1 (AMT_SIDE_LEFT); asynchronous on the domain's stream.  A sweep whose exchange
 * does not deliver then computes NaN in its boundary cells. */
enum amt_sides { AMT_SIDE_BELOW = 1, AMT_SIDE_ABOVE = 2, AMT_SIDE_LEFT = 4, AMT_SIDE_RIGHT = 8 };
int amt_domain_poison_halos(amt_domain *d, int sides);
/* enqueue n_sweeps calls of advance_mu_t over the patch's tile; asynchronous */
int amt_domain_step(amt_domain *d, int n_sweeps);
/* same, bracketed by HIP events on the domain's stream; returns after completion */
int amt_domain_step_timed(amt_domain *d, int n_sweeps, float *ms_total);
/* Placement tuning (speed only; +-3 % of a sweep hang on which physical pages the driver hands out):
 * allocates the handle's 26 arrays `tries` times, one set after the other (needs room for a second copy of the state;
 * stops early when there is none), copies the current contents over, times two sweeps of the handle's own kernel on each
 * set and keeps the fastest; every array holds afterwards what it held before the call.  ms_per_try (tries floats, or
 * NULL) receives the sweep time of each set (0 for sets that were not tried).  Device pointers obtained from
 * amt_domain_field_ptr before the call are no longer valid.  Not for wrapped domains (amt_domain_wrap). */
int amt_domain_tune_placement(amt_domain *domain, int tries, float *ms_per_try);
/* amt_domain_create does the same sampling by itself for states of 256 MiB and more (AMT_DOMAIN_PLACEMENT_TRIES allocations, default
 * 4; 0 or 1 = the first allocation as it comes; skipped where a second copy of the state does not fit): a host that creates its
 * handle once gets the sweep time bench.py prints.  COST, paid once at creation: up to TWICE the state plus a spacer of up to
 * 4 GiB resident for a moment, and 3 sweeps per allocation, timed with the AUTO kernel on the arrays as allocated (before
 * amt_domain_set_variant / the first upload: the placement of pages does not depend on either).  One process samples per device
 * at a time (advisory lock /tmp/amt_placement_<pci address>.lock); it is skipped altogether when the ranks of the launch
 * outnumber the visible devices (LOCAL_WORLD_SIZE, else WORLD_SIZE, > device count: ranks that share a device would sample beside
 * each other's sweeps and could starve each other's allocations).  Returns the number of allocations timed by the last sampling of this handle
 * (0: none) and their sweep times (ms; 0 for sets that were not tried). */
int amt_domain_placement(const amt_domain *domain, float *ms_per_try, int cap);
int amt_domain_sync(amt_domain *d);
void *amt_domain_field_ptr(amt_domain *d, int field);   /* device pointer, NULL on error */
void *amt_domain_stream(amt_domain *d);                 /* hipStream_t */

/* ------------------------------------------------------------------------
 * (4) Synthetic inputs (amt_synth.h) for a patch of extents idim x kdim x jdim
 *     whose element (0,0,0) is GLOBAL (gi0,gk0,gj0).  Rank-2 fields ignore the k
 *     arguments, rank-1 fields the i and j arguments.  The host and the device
 *     version produce identical bits.
 * ------------------------------------------------------------------------ */
int amt_synth_fill_host(int field, int dtype_bytes, void *dst, uint64_t seed,
                        long idim, long kdim, long jdim,
                        long gi0, long gk0, long gj0,
                        long gidim, long gkdim, long gjdim);
int amt_synth_fill_device(void *hip_stream, int field, int dtype_bytes, void *dst_device,
                          uint64_t seed,
                          long idim, long kdim, long jdim,
                          long gi0, long gk0, long gj0,
                          long gidim, long gkdim, long gjdim);

/* ------------------------------------------------------------------------
 * (5) j-slabs over several GPUs, one process per GPU: each rank owns rows jts..jte of the
 *     domain in a resident handle whose memory holds exactly one more row on either side
 *     (GLOBAL ids..jde, LOCAL jms = jts-1, jme = jte+1), as a WRF patch does.  Before a
 *     boundary row is computed the rank receives from the rank above row jte+1 of v, v_1, t_1,
 *     muv, msfvx_inv (module_small_step_em.f90:143-144,241) and from the rank below row jts-1
 *     of t_1 (:242): RCCL send/recv in one group on a communication stream, on which the two
 *     boundary rows then run, while the interior rows compute on the domain's stream.  Outputs
 *     need no exchange.  The reference splits j over its GPUs inside one process and refills
 *     the halos from the host on every call (advance_mu_t_no_async.cu:108-162).
 *     Two transports carry the rows.  RCCL (the default; loaded on first use with dlopen, AMT_RCCL_LIBRARY overrides the
 *     name): xGMI between the GPUs of a node; it needs one device per rank.  IPC (AMT_SLAB_TRANSPORT_IPC, or
 *     AMT_SLAB_TRANSPORT=ipc in the environment of a host that cannot pass the flag): no RCCL at all -- at creation the
 *     ranks trade hipIpcMemHandles of small staging buffers (copies of their boundary rows, refreshed by a kernel per sweep: the
 *     caller's arrays may be of any size and from any allocator) through a POSIX shared-memory block named after the unique id;
 *     per sweep the receiver waits for its neighbour's "rows final" number in that block (a one-wave kernel), pulls the rows
 *     with the copy engine (hipMemcpyAsync from the peer mapping: SDMA over xGMI between GPUs, no compute unit held while
 *     the wire is busy -- RCCL's send/recv kernel holds 31) and posts "pulled"; a rank's sweep ends when its neighbours have
 *     pulled its rows.  The ranks must be processes of one node; they may share ONE device (how the two-rank tests run on
 *     a one-GPU box).
 *     With overlap the IPC transport runs the HOST-WAITED schedule by default (AMT_IPC_HOST_WAIT=0: a waiting kernel instead): "rows
 *     final" is posted on the domain's stream in front of the interior, amt_slab_step waits on the calling host thread until the
 *     neighbours have posted theirs (so the call blocks for as long as a neighbour is behind -- per sub-step, like an MPI host),
 *     and the pull and the boundary rows run behind an interior that had the whole chip: nothing holds a compute unit while a
 *     neighbour is late (one rank of 8 in loopback: bare launch + 2.3 %, flat to 1.5 ms of lateness; profiles/r05_slab_ab.md).
 *     AMT_IPC_PULL=engine|kernel overrides how the rows are pulled (amt_slab_pull_mode; AMT_IPC_PULL_WGS: workgroups of the
 *     fused kernel, 4); AMT_IPC_TIMEOUT_S (120) bounds
 *     the host-side waits of the set-up, AMT_IPC_DEVICE_TIMEOUT_S (30) a device-side wait for a neighbour: it gives up,
 *     the sweep completes with invalid halo rows and amt_slab_sync returns AMT_ERR_COMM (nothing ever hangs the GPU).
 * ------------------------------------------------------------------------ */
#define AMT_UNIQUE_ID_BYTES 128              /* sizeof(ncclUniqueId) */
enum amt_slab_flags {
    AMT_SLAB_NO_OVERLAP = 1,                 /* exchange, then all rows, on one stream          */
    AMT_SLAB_LOOPBACK = 2,                   /* one-rank test mode: both neighbours are this rank */
    AMT_SLAB_TRANSPORT_IPC = 4               /* peer copies between processes instead of RCCL     */
};
typedef struct amt_slab amt_slab;

int amt_set_device(int device);              /* hipSetDevice for hosts without a HIP binding     */
/* rank 0: a fresh communicator id to hand to every rank: ncclGetUniqueId; where RCCL cannot be loaded, or with
 * AMT_SLAB_TRANSPORT=ipc in the environment, random bytes that name the launch for the IPC transport only */
int amt_comm_unique_id(void *id_out /* AMT_UNIQUE_ID_BYTES */);
/* The same through a file for hosts without MPI.  Rank 0 creates the id and publishes it as
 * `path` behind a header carrying `nonce`; every other rank waits (up to timeout_s) for a file
 * with ITS nonce, reads the id and acknowledges; rank 0 returns when all world-1 ranks have it and
 * removes the files.  `nonce` ties the file to one launch: a file left by another (crashed)
 * launch is never accepted, however fresh.  nonce = 0 means amt_comm_launch_nonce(). */
int amt_comm_rendezvous_file(const char *path, uint64_t nonce, int rank, int world,
                             double timeout_s, void *id_out);
/* a value all processes of one launch agree on and two launches do not: AMT_RENDEZVOUS_NONCE if
 * set, else the launcher (TORCHELASTIC_RUN_ID, parent pid + its start time) and MASTER_PORT */
uint64_t amt_comm_launch_nonce(void);
/* collective over the `world` ranks (ncclCommInitRank / the IPC set-up); the domain must outlive the slab */
int amt_slab_create(amt_slab **out, amt_domain *domain, int rank, int world,
                    const void *unique_id, int flags);
int amt_slab_destroy(amt_slab *slab);
int amt_slab_exchange(amt_slab *slab);       /* the halo exchange alone                           */
int amt_slab_step(amt_slab *slab, int n_sweeps);             /* asynchronous                      */
int amt_slab_step_timed(amt_slab *slab, int n_sweeps, float *ms_total);
int amt_slab_sync(amt_slab *slab);               /* AMT_ERR_COMM if a device-side wait for a neighbour gave up (IPC) */
const char *amt_slab_transport(const amt_slab *slab);        /* "rccl", "ipc", or "none" (a world of one)  */
/* how the IPC transport pulls: "copy engine" (between GPUs: hipMemcpyAsync per row) or "fused kernel" (ranks that share a device,
 * loopback: one kernel waits, pulls and posts -- a peer copy is a blit kernel there anyway); "" with RCCL */
const char *amt_slab_pull_mode(const amt_slab *slab);
/* Test hook: from now on every sweep's exchange starts `microseconds` late on the communication stream (a device-side
 * delay in front of the ncclSend/ncclRecv group), i.e. the neighbours' rows arrive that much late -- neighbour skew on
 * one GPU with the rank as its own neighbour (AMT_SLAB_LOOPBACK; profiles/slab_loopback.py --skew-us).  0 = off.  (The IPC
 * transport carries the delay inside its own waiting kernel: what waits is what would wait for a late neighbour.)
 * AMT_SLAB_SKEW_WGS=n in the environment gives the delay n workgroups that each hold a compute unit (31: what RCCL's
 * waiting send/recv kernel holds; default 1). */
int amt_slab_set_skew_us(amt_slab *slab, int microseconds);
long amt_slab_halo_bytes(const amt_slab *slab);              /* sent + received by this rank per sweep */
/* rank and size as the transport reports them: the communicator, or the ranks attached to the IPC block (0 of 1 without one) */
int amt_slab_comm_info(const amt_slab *slab, int *rank, int *world);
/* reporting aids for hosts without MPI: drain this rank's streams, then wait for every rank
 * (barrier) / replace *x by its maximum over the ranks.  The sweep itself has no collective. */
int amt_slab_barrier(amt_slab *slab);
int amt_slab_max(amt_slab *slab, double *x);

/* ------------------------------------------------------------------------
 * (5b) Patches in i AND j (domains whose j extent is too small for one slab per GPU): rank
 *      rj * pi + ri owns patch (ri, rj) of a pi x pj decomposition -- columns its..ite and rows jts..jte of the domain in a
 *      resident handle whose memory holds at least one more column and row on every side that has a neighbour (GLOBAL ids..jde,
 *      LOCAL ims <= its-1, ime >= ite+1, jms = jts-1, jme = jte+1).  Rows cross a boundary as in (5).  Across an i boundary the
 *      stencil reads column ite+1 of u, u_1, t_1, muu, msfuy (module_small_step_em.f90:145-146, :244) and column its-1 of t_1
 *      (:245): a column is kdim*jdim elements at stride idim, so one HIP kernel gathers the columns a patch sends into one
 *      contiguous buffer per direction, the buffers travel in the same exchange as the rows (same transports, same flags), and
 *      one kernel scatters what arrived into the halo columns.  No diagonal neighbours are needed.  Interior cells compute on
 *      the domain's stream beside the exchange; the boundary rows (over the patch's whole width: they own the corners) and the
 *      boundary columns (over the rows in between) follow it on the communication stream.  amt_slab_* is the pi = 1 case of the
 *      same stepper.  Flags are enum amt_slab_flags (AMT_SLAB_LOOPBACK: the rank is its own neighbour on all four sides).
 *      Cost on one patch (profiles/r05_grid_loopback.md): overlap +8 % at 2048 x 2048 columns, +20 % at 1024 x 1024 (three boundary
 *      launches per sweep); AMT_SLAB_NO_OVERLAP (halos, then ONE launch) +3 % and +7 % plus the neighbours' lateness.
 * ------------------------------------------------------------------------ */
typedef struct amt_grid amt_grid;
int amt_grid_create(amt_grid **out, amt_domain *domain, int ri, int rj, int pi, int pj,
                    const void *unique_id, int flags);               /* collective over the pi * pj ranks */
int amt_grid_destroy(amt_grid *grid);
int amt_grid_exchange(amt_grid *grid);                               /* pack, exchange, unpack alone      */
int amt_grid_step(amt_grid *grid, int n_sweeps);                     /* asynchronous                      */
int amt_grid_step_timed(amt_grid *grid, int n_sweeps, float *ms_total);
int amt_grid_sync(amt_grid *grid);
int amt_grid_set_skew_us(amt_grid *grid, int microseconds);          /* test hook, as amt_slab_set_skew_us */
long amt_grid_halo_bytes(const amt_grid *grid);                      /* sent + received by this rank per sweep */
const char *amt_grid_transport(const amt_grid *grid);
const char *amt_grid_pull_mode(const amt_grid *grid);
int amt_grid_comm_info(const amt_grid *grid, int *rank, int *world);
int amt_grid_barrier(amt_grid *grid);
int amt_grid_max(amt_grid *grid, double *x);

/* ------------------------------------------------------------------------
 * (6) Profiling aid: a plain streaming copy of nbytes (device to device) that moves
 *     bytes_per_lane = 4, 8 or 16 bytes per lane per access -- a KNOWN byte count in
 *     the kernels' own access width, used to calibrate rocprofv3's FETCH_SIZE /
 *     WRITE_SIZE on gfx950 (profiles/README.md).
 * ------------------------------------------------------------------------ */
int amt_calib_stream_copy(void *hip_stream, void *dst_device, const void *src_device,
                          size_t nbytes, int bytes_per_lane);
/* The box's own streaming ceilings, for attributing a measured sweep time to the machine or to the
 * kernel (bench.py: roofline.box_copy_GBps / box_read_GBps, measured in the same process): mode 0 copies
 * nbytes from src to dst (nbytes read + nbytes written), mode 1 only reads src (dst: 8 bytes that are never
 * written for finite data).  16 bytes per lane, tuned for gfx950; nbytes a multiple of 16.  Asynchronous:
 * time it with events on hip_stream. */
int amt_calib_stream_rate(void *hip_stream, void *dst_device, const void *src_device,
                          size_t nbytes, int mode);

/* ------------------------------------------------------------------------
 * (7) Tuning and test hooks of AMT_VARIANT_MARCH: force the wave shape
 *     (columns per lane, levels per lane, level groups per wave, extra DMA'd inputs, DMA or
 *     register flavour, rows per workgroup, 16- or 11-wave build; 0 / -1 = leave it to the launcher) for every later
 *     call of the process -- a shape that cannot run the given level count makes the call fail
 *     (AMT_VARIANT_AUTO then falls back to the column kernel).  amt_march_last_kernel names the
 *     kernel the calling thread's last launch plan chose (the column kernel included); amt_march_selectable lists the ones the
 *     launcher can choose unforced, one per line (returns the bytes needed).
 * ------------------------------------------------------------------------ */
int amt_march_force_shape(int vw, int kpt, int hl, int xd, int dma, int jrows, int max_waves);
const char *amt_march_last_kernel(void);
int amt_march_selectable(char *buf, int cap);
/* Rows per workgroup the launcher gives `ntile_i` column tiles of `nj` rows on `cus` compute units when a block's
 * unsigned 32-bit row offsets span at most `max_rows` rows, for a shape of `wbytes`-byte elements with `hl` level groups
 * per wave: the r <= max_rows that minimises rounds(r) * (r + 0.5), at most 64 for the fp64 shapes with level groups
 * (DESIGN.md section 4.2 "Launch plan", profiles/r04_rows.md).  Host arithmetic only; 0 when there is nothing to plan (no tile, row or CU). */
int amt_march_rows_for(long ntile_i, int nj, int cus, long max_rows, int wbytes, int hl);
/* Diagnosis: how workgroup numbers map to blocks.  0 (default): each XCD owns one contiguous run of blocks for the
 * whole launch; n > 0: every 8 n consecutive workgroup numbers cover 8 n consecutive blocks, n per XCD.  Which is
 * faster depends on where the arrays lie (profiles/r04_rows.md): not a tuning default. */
int amt_march_set_xchunk(int n);
/* How a launch made BESIDE another stream's kernels is planned (AMT_LAUNCH_BESIDE_OTHERS; a j-slab's interior).  A march workgroup
 * takes a compute unit whole (all its LDS and registers), so the other stream's kernels -- the halo exchange, the edge rows -- start
 * only where a workgroup ends, and every unit they take pushes interior workgroups into one more round.  `rounds`: least number of
 * rounds of workgroups (more, shorter rounds: more places to start, and a shorter extra round; half a row of prologue per block);
 * `reserve_cus`: compute units every round is planned to leave free (the other stream's kernels then start at once and push
 * nothing).  0, 0 = the defaults (AMT_MARCH_BESIDE_ROUNDS / AMT_MARCH_BESIDE_RESERVE in the environment, else 2 and 0).
 * Measured: profiles/r05_slab_ab.md. */
int amt_march_set_beside(int rounds, int reserve_cus);
/* Cache policy of the three once-read streams t, ft, ww_1 (same bits either way): -1 (default) by the
 * row length -- non-temporal loads (kernel amt_march_kernel) where rows are whole 128-byte lines, plain loads (kernel
 * amt_march_kernel_cached) where they are not (WRF's own unpadded ims:ime: neighbouring tiles then share the edge line of those
 * streams, and nt would drop it from L2: +3.5 % HBM reads, profiles/r06_rows4098_nt.md); 0 = always plain, 1 = always non-temporal.
 * AMT_MARCH_NT in the environment sets the same at load time. */
int amt_march_set_stream_policy(int policy);

#ifdef __cplusplus
}
#endif

#endif /* AMT_ADVANCE_MU_T_H */
