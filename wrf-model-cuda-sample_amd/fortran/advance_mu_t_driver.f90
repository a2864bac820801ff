! advance_mu_t_driver.f90 -- Fortran-90 host driver of the MI355X advance_mu_t path.
!
! Mirrors the flow of the reference driver (advance_mu_t_driver.f90: read the
! dimensions and the 26 arrays, time one CALL advance_mu_t, compare the 8
! outputs), with two differences: the inputs are the seeded synthetic fields of
! include/amt_synth.h (the reference's /data2/... dump is not shipped), and the
! CALL goes to the drop-in module_small_step_em of this directory, i.e. through
! ISO_C_BINDING into the HIP library.  It then repeats the sweep on the resident
! domain handle (device arrays kept across calls -- the kernel-only time the
! reference reports) and checks that both paths agree bit for bit.
!
!   advance_mu_t_driver [NI NK NJ [nsweeps [outdir [flags [placements]]]]]
!     flags: 0 none, 1 specified, 2 nested, 3 specified+periodic_x
!     placements: > 1 samples that many allocations of the resident state and keeps the fastest (amt_domain_tune_placement)
!   With outdir the 7 updated arrays are written there as raw native-endian
!   streams <name>.bin for an external checker.
program advance_mu_t_driver
  use iso_c_binding
  use amt_c_binding
  use module_configure, only : grid_config_rec_type
  use module_small_step_em, only : advance_mu_t
  implicit none

  integer, parameter :: wp = kind(1.0)          ! default REAL: fp32, or fp64 with -fdefault-real-8
  integer :: ni, nk, nj, nsweeps, iflag
  integer :: ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte
  character(len=256) :: arg, outdir
  type(grid_config_rec_type) :: config_flags
  real(wp), allocatable, target, dimension(:,:,:) :: ww, ww_1, u, u_1, v, v_1, t, t_1, t_ave, ft
  real(wp), allocatable, target, dimension(:,:,:) :: ww_r, t_r, t_ave_r
  real(wp), allocatable, target, dimension(:,:)   :: mu, mut, muave, muts, muu, muv, mudf, mu_tend
  real(wp), allocatable, target, dimension(:,:)   :: msfuy, msfvx_inv, msftx, msfty
  real(wp), allocatable, target, dimension(:,:)   :: mu_r, muave_r, muts_r, mudf_r
  real(wp), allocatable, target, dimension(:)     :: dnw, fnm, fnp, rdnw
  real(wp) :: rdx, rdy, dts, epssm
  integer(c_int64_t), parameter :: seed = 12345_c_int64_t
  integer(c_int) :: rc
  type(c_ptr) :: dom
  real(c_float) :: ms
  integer(kind=8) :: c0, c1, cmid, hz
  real(kind=8) :: cells, secs
  integer :: nbad, ndef, ntune
  integer(c_int) :: nslots, slot_ids(16)
  real(c_float) :: tune_ms(8)

  ni = 64; nk = 40; nj = 64; nsweeps = 5; outdir = ' '; iflag = 0      ! BASELINE.json configs[0]
  if (command_argument_count() >= 3) then
     call get_command_argument(1, arg); read (arg, *) ni
     call get_command_argument(2, arg); read (arg, *) nk
     call get_command_argument(3, arg); read (arg, *) nj
  end if
  if (command_argument_count() >= 4) then
     call get_command_argument(4, arg); read (arg, *) nsweeps
  end if
  if (command_argument_count() >= 5) call get_command_argument(5, outdir)
  if (command_argument_count() >= 6) then
     call get_command_argument(6, arg); read (arg, *) iflag
  end if
  ntune = 0
  if (command_argument_count() >= 7) then
     call get_command_argument(7, arg); read (arg, *) ntune
     ntune = min(ntune, 8)
  end if
  config_flags%specified  = (iflag == 1 .or. iflag == 3)
  config_flags%nested     = (iflag == 2)
  config_flags%periodic_x = (iflag == 3)

  ! single-patch domain, SURVEY.md section 8 convention
  ids = 1; ide = ni + 1; jds = 1; jde = nj + 1; kde = nk + 1
  ims = 0; ime = ni + 1; jms = 0; jme = nj + 1; kms = 1; kme = nk + 1
  its = 1; ite = ide;    jts = 1; jte = jde;    kts = 1; kte = kde
  rdx = 1.0e-3_wp; rdy = 1.25e-3_wp; dts = 2.0_wp; epssm = 0.1_wp      ! AMT_SYNTH_* of amt_synth.h

  allocate (ww(ims:ime,kms:kme,jms:jme), ww_1(ims:ime,kms:kme,jms:jme), u(ims:ime,kms:kme,jms:jme))
  allocate (u_1(ims:ime,kms:kme,jms:jme), v(ims:ime,kms:kme,jms:jme), v_1(ims:ime,kms:kme,jms:jme))
  allocate (t(ims:ime,kms:kme,jms:jme), t_1(ims:ime,kms:kme,jms:jme), t_ave(ims:ime,kms:kme,jms:jme))
  allocate (ft(ims:ime,kms:kme,jms:jme))
  allocate (ww_r(ims:ime,kms:kme,jms:jme), t_r(ims:ime,kms:kme,jms:jme), t_ave_r(ims:ime,kms:kme,jms:jme))
  allocate (mu(ims:ime,jms:jme), mut(ims:ime,jms:jme), muave(ims:ime,jms:jme), muts(ims:ime,jms:jme))
  allocate (muu(ims:ime,jms:jme), muv(ims:ime,jms:jme), mudf(ims:ime,jms:jme), mu_tend(ims:ime,jms:jme))
  allocate (msfuy(ims:ime,jms:jme), msfvx_inv(ims:ime,jms:jme), msftx(ims:ime,jms:jme), msfty(ims:ime,jms:jme))
  allocate (mu_r(ims:ime,jms:jme), muave_r(ims:ime,jms:jme), muts_r(ims:ime,jms:jme), mudf_r(ims:ime,jms:jme))
  allocate (dnw(kms:kme), fnm(kms:kme), fnp(kms:kme), rdnw(kms:kme))

  call fill3(AMT_F_WW, ww);   call fill3(AMT_F_WW_1, ww_1); call fill3(AMT_F_U, u);   call fill3(AMT_F_U_1, u_1)
  call fill3(AMT_F_V, v);     call fill3(AMT_F_V_1, v_1);   call fill3(AMT_F_T, t);   call fill3(AMT_F_T_1, t_1)
  call fill3(AMT_F_T_AVE, t_ave); call fill3(AMT_F_FT, ft)
  call fill2(AMT_F_MU, mu);   call fill2(AMT_F_MUT, mut);   call fill2(AMT_F_MUAVE, muave); call fill2(AMT_F_MUTS, muts)
  call fill2(AMT_F_MUU, muu); call fill2(AMT_F_MUV, muv);   call fill2(AMT_F_MUDF, mudf);   call fill2(AMT_F_MU_TEND, mu_tend)
  call fill2(AMT_F_MSFUY, msfuy); call fill2(AMT_F_MSFVX_INV, msfvx_inv)
  call fill2(AMT_F_MSFTX, msftx); call fill2(AMT_F_MSFTY, msfty)
  call fill1(AMT_F_DNW, dnw); call fill1(AMT_F_FNM, fnm);   call fill1(AMT_F_FNP, fnp);     call fill1(AMT_F_RDNW, rdnw)

  print '(a,i0,a,i0,a,i0,a,i0,a,i0)', 'advance_mu_t ', ni, 'x', nk, 'x', nj, ' real*', storage_size(rdx)/8, &
        '  HIP devices: ', amt_device_count()

  ! ---- resident path first (it needs the un-updated inputs) ----
  rc = amt_domain_create(dom, int(storage_size(rdx)/8, c_int),                                 &
                         merge(1_c_int, 0_c_int, config_flags%periodic_x),                      &
                         merge(1_c_int, 0_c_int, config_flags%specified),                       &
                         merge(1_c_int, 0_c_int, config_flags%nested),                          &
                         ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte)
  call amt_check(rc, 'amt_domain_create')
  call amt_check(amt_domain_set_scalars(dom, real(rdx, c_double), real(rdy, c_double), real(dts, c_double), &
                                        real(epssm, c_double)), 'amt_domain_set_scalars')
  ! one untimed sweep on whatever the fresh buffers hold: loads the kernel's code object, so that the
  ! timed sweeps below are kernel time only; every array is uploaded after it
  call amt_check(amt_domain_step(dom, 1_c_int), 'amt_domain_step (warm-up)')
  call amt_check(amt_domain_sync(dom), 'amt_domain_sync')
  call up(AMT_F_WW, c_loc(ww));     call up(AMT_F_WW_1, c_loc(ww_1)); call up(AMT_F_U, c_loc(u))
  call up(AMT_F_U_1, c_loc(u_1));   call up(AMT_F_V, c_loc(v));       call up(AMT_F_V_1, c_loc(v_1))
  call up(AMT_F_T, c_loc(t));       call up(AMT_F_T_1, c_loc(t_1));   call up(AMT_F_T_AVE, c_loc(t_ave))
  call up(AMT_F_FT, c_loc(ft));     call up(AMT_F_MU, c_loc(mu));     call up(AMT_F_MUT, c_loc(mut))
  call up(AMT_F_MUAVE, c_loc(muave)); call up(AMT_F_MUTS, c_loc(muts)); call up(AMT_F_MUU, c_loc(muu))
  call up(AMT_F_MUV, c_loc(muv));   call up(AMT_F_MUDF, c_loc(mudf)); call up(AMT_F_MU_TEND, c_loc(mu_tend))
  call up(AMT_F_MSFUY, c_loc(msfuy)); call up(AMT_F_MSFVX_INV, c_loc(msfvx_inv))
  call up(AMT_F_MSFTX, c_loc(msftx)); call up(AMT_F_MSFTY, c_loc(msfty))
  call up(AMT_F_DNW, c_loc(dnw));   call up(AMT_F_FNM, c_loc(fnm));   call up(AMT_F_FNP, c_loc(fnp))
  call up(AMT_F_RDNW, c_loc(rdnw))
  ! placement of the arrays' pages (+-3 % of a sweep, DESIGN.md section 4.3): sampled by the library, contents kept
  if (ntune > 1) then
     call amt_check(amt_domain_tune_placement(dom, int(ntune, c_int), tune_ms), 'amt_domain_tune_placement')
     print '(a,i0,a,8f9.4)', 'placement tuning, ', ntune, ' allocations, ms/sweep each: ', tune_ms(1:ntune)
  end if
  call amt_check(amt_domain_step_timed(dom, int(nsweeps, c_int), ms), 'amt_domain_step_timed')
  cells = real(ni, 8) * real(nk, 8) * real(nj, 8)
  print '(a,i0,a,f10.4,a,f12.1,a)', 'resident device path: ', nsweeps, ' sweeps, ', ms / nsweeps, &
        ' ms/sweep (kernel only), ', cells * nsweeps / (ms * 1.0d-3) / 1.0d6, ' Mcells/s'
  call dn(AMT_F_WW, c_loc(ww_r));   call dn(AMT_F_T, c_loc(t_r));     call dn(AMT_F_T_AVE, c_loc(t_ave_r))
  call dn(AMT_F_MU, c_loc(mu_r));   call dn(AMT_F_MUAVE, c_loc(muave_r)); call dn(AMT_F_MUTS, c_loc(muts_r))
  call dn(AMT_F_MUDF, c_loc(mudf_r))
  call amt_check(amt_domain_destroy(dom), 'amt_domain_destroy')

  ! ---- one-shot drop-in: the reference driver's CALL (advance_mu_t_driver.f90:193-205), nsweeps times ----
  call system_clock(count_rate=hz)
  call system_clock(count=c0)
  block
    integer :: s
    do s = 1, nsweeps
      CALL advance_mu_t( ww, ww_1, u, u_1, v, v_1,            &
                         mu, mut, muave, muts, muu, muv,      &
                         mudf, t, t_1,                        &
                         t_ave, ft, mu_tend,                  &
                         rdx, rdy, dts, epssm,                &
                         dnw, fnm, fnp, rdnw,                 &
                         msfuy, msfvx_inv,                    &
                         msftx, msfty,                        &
                         config_flags,                        &
                         ids, ide, jds, jde, kde,             &
                         ims, ime, jms, jme, kms, kme,        &
                         its, ite, jts, jte, kts, kte )
      if (s == 1) call system_clock(count=cmid)
    end do
  end block
  call system_clock(count=c1)
  ! the first call creates this thread's device workspace (streams, events, arena), which later calls reuse
  secs = real(cmid - c0, 8) / real(hz, 8)
  print '(a,f10.4,a)', 'one-shot host path:   first call ', secs * 1.0d3, ' ms (creates the device workspace)'
  nslots = amt_host_devices(slot_ids, 16_c_int)             ! AMT_ONESHOT_DEVICES="0,1,..." | "all": one call, several devices
  if (nslots > 0) print '(a,i0,a,16(1x,i0))', 'one-shot host path:   every call fans its rows over ', nslots, ' device slot(s):', slot_ids(1:nslots)
  if (nsweeps > 1) then
     secs = real(c1 - cmid, 8) / real(hz, 8)
     print '(a,i0,a,f10.4,a,f12.1,a)', 'one-shot host path:   ', nsweeps - 1, ' calls,  ', secs * 1.0d3 / (nsweeps - 1), &
           ' ms/call  (H2D+kernel+D2H), ', cells * (nsweeps - 1) / secs / 1.0d6, ' Mcells/s'
  end if

  ! ---- the same calls with the ten 3-D arrays page-locked once (the reference driver allocates its
  !      host buffers pinned, advance_mu_t_driver.cu:97-167): the library then streams the window in
  !      j chunks (H2D / kernel / D2H overlapped on three streams).  Timing only: the arrays have
  !      already been advanced nsweeps times, these extra sweeps are undone by nothing and the
  !      comparison below is therefore made BEFORE them.
  nbad = 0
  nbad = nbad + count(transfer(ww, 1_1, size(ww) * storage_size(rdx) / 8) /= transfer(ww_r, 1_1, size(ww) * storage_size(rdx) / 8))
  nbad = nbad + count(t /= t_r) + count(t_ave /= t_ave_r) + count(mu /= mu_r)
  nbad = nbad + count(muave /= muave_r) + count(muts /= muts_r) + count(mudf /= mudf_r)
  if (len_trim(outdir) > 0) then
     call dump3('ww', ww); call dump3('t', t); call dump3('t_ave', t_ave)
     call dump2('mu', mu); call dump2('muave', muave); call dump2('muts', muts); call dump2('mudf', mudf)
  end if
  print '(a,4es24.16)', 'checksums ww t mu muave: ', sum(real(ww, 8)), sum(real(t, 8)), sum(real(mu, 8)), sum(real(muave, 8))
  call pin3(ww); call pin3(ww_1); call pin3(u); call pin3(u_1); call pin3(v); call pin3(v_1)
  call pin3(t); call pin3(t_1); call pin3(t_ave); call pin3(ft)
  call system_clock(count=c0)
  block
    integer :: s
    do s = 1, nsweeps
      CALL advance_mu_t( ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,        &
                         t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,                 &
                         msfuy, msfvx_inv, msftx, msfty, config_flags,                                  &
                         ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte )
    end do
  end block
  call system_clock(count=c1)
  secs = real(c1 - c0, 8) / real(hz, 8)
  print '(a,i0,a,f10.4,a,f12.1,a)', 'one-shot, pinned host: ', nsweeps, ' calls, ', secs * 1.0d3 / nsweeps, &
        ' ms/call  (streamed in j chunks),  ', cells * nsweeps / secs / 1.0d6, ' Mcells/s'
  ! ---- the acoustic loop of one Runge-Kutta stage: the linearisation state ww_1, u_1, v_1, t_1 and the tendency
  !      ft do not change between the sub-steps, so with the residency cache on they are uploaded by the first call
  !      only (amt_host_invalidate(c_null_ptr) when the next stage has rewritten them).  Timing only, as above.
  ww_r = ww; t_r = t; mu_r = mu                       ! the state this loop starts from (the deferred loop below repeats it)
  call amt_check(amt_host_cache_enable(1_c_int), 'amt_host_cache_enable')
  CALL advance_mu_t( ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,            &
                     t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,                     &
                     msfuy, msfvx_inv, msftx, msfty, config_flags,                                      &
                     ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte )
  call system_clock(count=c0)
  block
    integer :: s
    do s = 1, nsweeps
      CALL advance_mu_t( ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,        &
                         t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,                 &
                         msfuy, msfvx_inv, msftx, msfty, config_flags,                                  &
                         ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte )
    end do
  end block
  call system_clock(count=c1)
  secs = real(c1 - c0, 8) / real(hz, 8)
  print '(a,i0,a,f10.4,a,f12.1,a)', 'one-shot, pinned, constants resident: ', nsweeps, ' calls, ', secs * 1.0d3 / nsweeps, &
        ' ms/call,  ', cells * nsweeps / secs / 1.0d6, ' Mcells/s'
  ! ---- the same loop with the OUTPUTS deferred as well (amt_host_defer): ww, t, t_ave, mu, muave, muts, mudf stay on the
  !      device from sub-step to sub-step -- in WRF the next routine of the acoustic loop would take them there -- and come
  !      down once, when the loop is over (amt_host_fetch); a sub-step then uploads u, v and the 2-D / 1-D inputs only.
  !      Starts from the state the loop above started from and must end with its bits.
  call swap3(ww, ww_r); call swap3(t, t_r); call swap2(mu, mu_r)
  t_ave_r = t_ave; muave_r = muave; muts_r = muts; mudf_r = mudf
  call amt_check(amt_host_defer(c_null_ptr, 1_c_int), 'amt_host_defer')
  CALL advance_mu_t( ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,            &
                     t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,                     &
                     msfuy, msfvx_inv, msftx, msfty, config_flags,                                      &
                     ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte )
  call system_clock(count=c0)
  block
    integer :: s
    do s = 1, nsweeps
      CALL advance_mu_t( ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv, mudf, t, t_1,        &
                         t_ave, ft, mu_tend, rdx, rdy, dts, epssm, dnw, fnm, fnp, rdnw,                 &
                         msfuy, msfvx_inv, msftx, msfty, config_flags,                                  &
                         ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte )
    end do
  end block
  call system_clock(count=c1)
  secs = real(c1 - c0, 8) / real(hz, 8)
  print '(a,i0,a,f10.4,a,f12.1,a)', 'one-shot, pinned, constants resident, outputs deferred: ', nsweeps, ' calls, ', &
        secs * 1.0d3 / nsweeps, ' ms/call,  ', cells * nsweeps / secs / 1.0d6, ' Mcells/s'
  if (amt_host_stale(c_loc(t)) /= 1) error stop 3
  call system_clock(count=c0)
  call amt_check(amt_host_fetch(c_null_ptr), 'amt_host_fetch')
  call system_clock(count=c1)
  print '(a,f10.4,a)', 'amt_host_fetch of the seven outputs, once per loop: ', real(c1 - c0, 8) / real(hz, 8) * 1.0d3, ' ms'
  if (amt_host_stale(c_null_ptr) /= 0) error stop 4
  call amt_check(amt_host_defer(c_null_ptr, 0_c_int), 'amt_host_defer')
  ndef = 0
  ndef = ndef + count(transfer(ww, 1_1, size(ww) * storage_size(rdx) / 8) /= transfer(ww_r, 1_1, size(ww) * storage_size(rdx) / 8))
  ndef = ndef + count(t /= t_r) + count(t_ave /= t_ave_r) + count(mu /= mu_r)
  ndef = ndef + count(muave /= muave_r) + count(muts /= muts_r) + count(mudf /= mudf_r)
  print '(a,i0)', 'deferred loop vs plain loop: differing elements = ', ndef
  call amt_check(amt_host_cache_enable(0_c_int), 'amt_host_cache_enable')
  call unpin3(ww); call unpin3(ww_1); call unpin3(u); call unpin3(u_1); call unpin3(v); call unpin3(v_1)
  call unpin3(t); call unpin3(t_1); call unpin3(t_ave); call unpin3(ft)

  ! ---- both paths must agree bit for bit (compared above, before the pinned timing run) ----
  print '(a,i0)', 'one-shot vs resident: differing elements = ', nbad
  if (nbad /= 0) error stop 2
  if (ndef /= 0) error stop 5

contains

  subroutine fill3(field, a)
    integer(c_int), intent(in) :: field
    real(wp), target, intent(inout) :: a(ims:, kms:, jms:)
    call amt_check(amt_synth_fill_host(field, int(storage_size(rdx)/8, c_int), c_loc(a), seed,         &
         int(ime-ims+1, c_long), int(kme-kms+1, c_long), int(jme-jms+1, c_long),                        &
         int(ims, c_long), int(kms-1, c_long), int(jms, c_long),                                        &
         int(ni+2, c_long), int(nk+1, c_long), int(nj+2, c_long)), 'amt_synth_fill_host')
  end subroutine
  subroutine fill2(field, a)
    integer(c_int), intent(in) :: field
    real(wp), target, intent(inout) :: a(ims:, jms:)
    call amt_check(amt_synth_fill_host(field, int(storage_size(rdx)/8, c_int), c_loc(a), seed,         &
         int(ime-ims+1, c_long), 1_c_long, int(jme-jms+1, c_long),                                      &
         int(ims, c_long), 0_c_long, int(jms, c_long),                                                  &
         int(ni+2, c_long), int(nk+1, c_long), int(nj+2, c_long)), 'amt_synth_fill_host')
  end subroutine
  subroutine fill1(field, a)
    integer(c_int), intent(in) :: field
    real(wp), target, intent(inout) :: a(kms:)
    call amt_check(amt_synth_fill_host(field, int(storage_size(rdx)/8, c_int), c_loc(a), seed,         &
         1_c_long, int(kme-kms+1, c_long), 1_c_long, 0_c_long, int(kms-1, c_long), 0_c_long,            &
         int(ni+2, c_long), int(nk+1, c_long), int(nj+2, c_long)), 'amt_synth_fill_host')
  end subroutine
  subroutine up(field, host)
    integer(c_int), intent(in) :: field
    type(c_ptr), intent(in) :: host
    call amt_check(amt_domain_upload(dom, field, host), 'amt_domain_upload')
  end subroutine
  subroutine dn(field, host)
    integer(c_int), intent(in) :: field
    type(c_ptr), intent(in) :: host
    call amt_check(amt_domain_download(dom, field, host), 'amt_domain_download')
  end subroutine
  subroutine swap3(a, b)
    real(wp), intent(inout) :: a(:,:,:), b(:,:,:)
    real(wp) :: x
    integer :: i, k, j
    do j = 1, size(a, 3)
      do k = 1, size(a, 2)
        do i = 1, size(a, 1)
          x = a(i, k, j); a(i, k, j) = b(i, k, j); b(i, k, j) = x
        end do
      end do
    end do
  end subroutine
  subroutine swap2(a, b)
    real(wp), intent(inout) :: a(:,:), b(:,:)
    real(wp) :: x
    integer :: i, j
    do j = 1, size(a, 2)
      do i = 1, size(a, 1)
        x = a(i, j); a(i, j) = b(i, j); b(i, j) = x
      end do
    end do
  end subroutine
  subroutine pin3(a)
    real(wp), target, intent(inout) :: a(:,:,:)
    call amt_check(amt_host_pin(c_loc(a), int(size(a), c_size_t) * int(storage_size(rdx) / 8, c_size_t)), 'amt_host_pin')
  end subroutine
  subroutine unpin3(a)
    real(wp), target, intent(inout) :: a(:,:,:)
    call amt_check(amt_host_unpin(c_loc(a)), 'amt_host_unpin')
  end subroutine
  subroutine dump3(name, a)
    character(len=*), intent(in) :: name
    real(wp), intent(in) :: a(:,:,:)
    integer :: un
    open (newunit=un, file=trim(outdir)//'/'//name//'.bin', access='stream', form='unformatted', status='replace')
    write (un) a
    close (un)
  end subroutine
  subroutine dump2(name, a)
    character(len=*), intent(in) :: name
    real(wp), intent(in) :: a(:,:)
    integer :: un
    open (newunit=un, file=trim(outdir)//'/'//name//'.bin', access='stream', form='unformatted', status='replace')
    write (un) a
    close (un)
  end subroutine

end program advance_mu_t_driver
