! advance_mu_t_grid_driver.f90 -- Fortran-90 host of the run decomposed in i AND j: one process per GPU (or, with the IPC
! transport, several processes on one), patch (ri, rj) of PI x PJ per rank, rank = rj * PI + ri.
!
! Every rank owns columns ilo..ihi and rows jlo..jhi of the NI x NK x NJ domain as a resident device patch (GLOBAL ids..jde,
! LOCAL ims <= ilo-1, ime >= ihi+1, jms = jlo-1, jme = jhi+1 -- the triple WRF itself passes), filled from the seeded index-based
! generator of include/amt_synth.h, and calls amt_grid_step (include/amt_advance_mu_t.h section 5b): halo rows in place, halo
! columns of u, u_1, t_1, muu, msfuy gathered / scattered by HIP kernels, one exchange, interior cells beside it
! (the i+1 / i-1 reads of module_small_step_em.f90:145-146, 244-245; SURVEY.md section 8f row 4).
!
!   advance_mu_t_grid_driver NI NK NJ nsweeps PI PJ
!   RANK / WORLD_SIZE (= PI * PJ) / LOCAL_RANK / MASTER_PORT from the launcher; AMT_RENDEZVOUS_FILE as in the slab driver;
!   AMT_SLAB_TRANSPORT=ipc selects the RCCL-free transport (ranks may then share a device).
!   AMT_GRID_POISON=1   overwrite every halo row / column that has a neighbour with NaN before stepping (only a working exchange
!                       then gives finite results)
!   AMT_GRID_REFRESH=1  one amt_grid_step per sweep, and before every sweep but the first NEW values in the fields that cross a
!                       patch boundary (amt_domain_fill_fields, seed + sweep: the stand-in for advance_uv) and, with
!                       AMT_GRID_POISON, NaN in the halos again: only an exchange that delivers every sweep gives the right bits
!   AMT_GRID_DUMP_DIR   write the seven outputs of this rank (whole memory arrays, stream access) as <dir>/rank<r>_<name>.bin with
!                       <dir>/rank<r>_bounds.txt for a checker (tests/test_gpu_33_grid_native.py holds the unsplit oracle run)
program advance_mu_t_grid_driver
  use iso_c_binding
  use amt_c_binding
  implicit none

  integer, parameter :: wp = kind(1.0)          ! default REAL: fp32, or fp64 with -fdefault-real-8
  integer :: ni, nk, nj, nsweeps, pi, pj, ri, rj, rank, world, local_rank
  integer :: ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte
  integer :: ilo, ihi, jlo, jhi, idim, n, f
  character(len=256) :: arg
  character(len=512) :: path, port, dumpdir
  character(kind=c_char), target :: uid(128)
  character(kind=c_char), allocatable :: cpath(:)
  integer(c_int64_t), parameter :: seed = 12345_c_int64_t
  type(c_ptr) :: dom, grid, idptr
  real(c_float) :: ms, ms1
  real(wp), allocatable, target :: a3(:,:,:), a2(:,:)
  integer :: sides
  logical :: poison, refresh
  real(c_double) :: ms_job
  integer(c_int) :: crank, cworld
  integer(kind=8) :: c0, c1, crate
  integer, parameter :: outputs(7) = [AMT_F_WW, AMT_F_T, AMT_F_T_AVE, AMT_F_MU, AMT_F_MUAVE, AMT_F_MUTS, AMT_F_MUDF]
  character(len=8), parameter :: output_names(7) = [character(len=8) :: 'ww', 't', 't_ave', 'mu', 'muave', 'muts', 'mudf']

  if (command_argument_count() < 6) error stop 'usage: advance_mu_t_grid_driver NI NK NJ nsweeps PI PJ'
  call get_command_argument(1, arg); read (arg, *) ni
  call get_command_argument(2, arg); read (arg, *) nk
  call get_command_argument(3, arg); read (arg, *) nj
  call get_command_argument(4, arg); read (arg, *) nsweeps
  call get_command_argument(5, arg); read (arg, *) pi
  call get_command_argument(6, arg); read (arg, *) pj
  rank = env_int('RANK', 0)
  world = env_int('WORLD_SIZE', 1)
  local_rank = env_int('LOCAL_RANK', rank)
  if (world /= pi * pj) error stop 'WORLD_SIZE must be PI * PJ'
  if (ni < pi .or. nj < pj) error stop 'fewer columns or rows than patches'
  ri = mod(rank, pi); rj = rank / pi
  call amt_check(amt_set_device(int(local_rank, c_int)), 'amt_set_device')

  ! the domain (SURVEY.md section 8 convention) and this rank's patch of it: one halo column and row on every side, the i memory
  ! padded to whole 32-element runs with i = its on a run boundary
  ids = 1; ide = ni + 1; jds = 1; jde = nj + 1; kde = nk + 1
  ilo = ids + ((ide - ids) * ri) / pi
  ihi = ids + ((ide - ids) * (ri + 1)) / pi - 1
  jlo = jds + ((jde - jds) * rj) / pj
  jhi = jds + ((jde - jds) * (rj + 1)) / pj - 1
  ims = ilo - 32
  idim = ((ihi + 1 - ims + 1 + 31) / 32) * 32
  ime = ims + idim - 1
  kms = 1; kme = nk + 1; kts = 1; kte = kde
  its = ilo; ite = ihi; jts = jlo; jte = jhi; jms = jlo - 1; jme = jhi + 1

  call amt_check(amt_domain_create(dom, int(storage_size(1.0_wp)/8, c_int), 0_c_int, 0_c_int, 0_c_int,  &
                                   ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,            &
                                   its, ite, jts, jte, kts, kte), 'amt_domain_create')
  call amt_check(amt_domain_fill_synthetic(dom, seed, int(ims, c_long), int(kms - 1, c_long), int(jms, c_long), &
                                           int(ni + 2, c_long), int(nk + 1, c_long), int(nj + 2, c_long)), &
                 'amt_domain_fill_synthetic')
  allocate (a3(ims:ime, kms:kme, jms:jme), a2(ims:ime, jms:jme))

  ! the sides on which this patch reads a neighbour's rows / columns
  sides = 0
  if (rj > 0) sides = sides + AMT_SIDE_BELOW
  if (rj < pj - 1) sides = sides + AMT_SIDE_ABOVE
  if (ri > 0) sides = sides + AMT_SIDE_LEFT
  if (ri < pi - 1) sides = sides + AMT_SIDE_RIGHT
  poison = env_int('AMT_GRID_POISON', 0) /= 0
  refresh = env_int('AMT_GRID_REFRESH', 0) /= 0
  if (poison) call amt_check(amt_domain_poison_halos(dom, int(sides, c_int)), 'amt_domain_poison_halos')

  idptr = c_null_ptr
  if (world > 1) then
     call get_environment_variable('AMT_RENDEZVOUS_FILE', path)
     if (len_trim(path) == 0) then
        call get_environment_variable('MASTER_PORT', port)
        if (len_trim(port) == 0) port = '0'
        path = '/tmp/amt_uid_' // trim(port)
     end if
     allocate (cpath(len_trim(path) + 1))
     do n = 1, len_trim(path)
        cpath(n) = path(n:n)
     end do
     cpath(len_trim(path) + 1) = c_null_char
     call amt_check(amt_comm_rendezvous_file(cpath, 0_c_int64_t, int(rank, c_int), int(world, c_int), &
                                             120.0_c_double, uid), 'amt_comm_rendezvous_file')
     idptr = c_loc(uid)
  end if
  call amt_check(amt_grid_create(grid, dom, int(ri, c_int), int(rj, c_int), int(pi, c_int), int(pj, c_int), idptr, 0_c_int), &
                 'amt_grid_create')

  call amt_check(amt_grid_barrier(grid), 'amt_grid_barrier')
  call system_clock(c0, crate)
  if (refresh) then
     ! what a host model does: before every sub-step but the first the fields that cross a patch boundary get new values
     ! (amt_domain_fill_fields with seed + sweep stands in for advance_uv rewriting u, v), and -- for verification -- the halos
     ! are poisoned again: only an exchange that delivers THIS sweep's rows and columns gives the unsplit run's bits
     ms = 0.0
     do n = 0, nsweeps - 1
        if (n > 0) then
           call amt_check(amt_domain_fill_fields(dom, AMT_EXCHANGED_FIELDS, seed + int(n, c_int64_t), int(ims, c_long),    &
                                                 int(kms - 1, c_long), int(jms, c_long), int(ni + 2, c_long),            &
                                                 int(nk + 1, c_long), int(nj + 2, c_long)), 'amt_domain_fill_fields')
           if (poison) call amt_check(amt_domain_poison_halos(dom, int(sides, c_int)), 'amt_domain_poison_halos')
        end if
        call amt_check(amt_grid_step_timed(grid, 1_c_int, ms1), 'amt_grid_step_timed')
        ms = ms + ms1
     end do
  else
     call amt_check(amt_grid_step_timed(grid, int(nsweeps, c_int), ms), 'amt_grid_step_timed')
  end if
  call amt_check(amt_grid_sync(grid), 'amt_grid_sync')
  call system_clock(c1)
  ms_job = real(c1 - c0, 8) * 1.0d3 / real(crate, 8)
  call amt_check(amt_grid_max(grid, ms_job), 'amt_grid_max')
  call amt_check(amt_grid_comm_info(grid, crank, cworld), 'amt_grid_comm_info')

  call amt_check(amt_domain_download(dom, AMT_F_MU, c_loc(a2)), 'amt_domain_download')
  print '(a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,f9.4,a,i0,a,es22.14)',                                  &
        'rank ', rank, ' = patch (', ri, ',', rj, ') of ', pi, 'x', pj, ': i ', ilo, '..', ihi, ' j ', jlo, ' ', &
        ms / max(nsweeps, 1), ' ms/sweep; halo bytes/sweep ', amt_grid_halo_bytes(grid), '; sum(mu) ',       &
        sum(real(a2(ilo:ihi, jlo:jhi), 8))
  if (rank == 0) print '(a,i0,a,f9.4,a)', 'job: ', cworld, ' rank(s) seen by the transport, slowest rank ', &
        ms_job / max(nsweeps, 1), ' ms/sweep wall'

  call get_environment_variable('AMT_GRID_DUMP_DIR', dumpdir)
  if (len_trim(dumpdir) > 0) then
     open (unit=21, file=trim(dumpdir) // '/rank' // trim(itoa(rank)) // '_bounds.txt', status='replace')
     write (21, '(10(i0,1x))') ims, ime, kms, kme, jms, jme, ilo, ihi, jlo, jhi
     close (21)
     do n = 1, 7
        open (unit=22, file=trim(dumpdir) // '/rank' // trim(itoa(rank)) // '_' // trim(output_names(n)) // '.bin', &
              access='stream', form='unformatted', status='replace')
        if (n <= 3) then
           call amt_check(amt_domain_download(dom, outputs(n), c_loc(a3)), 'amt_domain_download')
           write (22) a3
        else
           call amt_check(amt_domain_download(dom, outputs(n), c_loc(a2)), 'amt_domain_download')
           write (22) a2
        end if
        close (22)
     end do
  end if

  call amt_check(amt_grid_destroy(grid), 'amt_grid_destroy')
  call amt_check(amt_domain_destroy(dom), 'amt_domain_destroy')

contains

  function itoa(v) result(s)
    integer, intent(in) :: v
    character(len=12) :: s
    write (s, '(i0)') v
  end function itoa

  integer function env_int(name, dflt) result(v)
    character(len=*), intent(in) :: name
    integer, intent(in) :: dflt
    character(len=64) :: s
    integer :: ios
    call get_environment_variable(name, s)
    v = dflt
    if (len_trim(s) > 0) then
       read (s, *, iostat=ios) v
       if (ios /= 0) v = dflt
    end if
  end function env_int

end program advance_mu_t_grid_driver
