! advance_mu_t_slab_driver.f90 -- Fortran-90 host of the j-decomposed run: one process per GPU.
!
! Every rank owns the rows jlo..jhi of the NI x NK x NJ domain as a resident device patch
! (GLOBAL ids..jde, LOCAL jms = jlo-1, jme = jhi+1 -- the triple WRF itself passes), filled from
! the seeded index-based generator of include/amt_synth.h, and calls amt_slab_step: the one-row
! input halos travel as RCCL send/recv on a communication stream together with the two boundary
! rows while the interior rows compute (include/amt_advance_mu_t.h section 5).  The reference
! splits j over its GPUs inside one process and refills the halos from the host on every call
! (advance_mu_t_no_async.cu:108-162).
!
!   RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT come from the launcher, e.g.
!     python -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node 8 \
!         --master-addr 127.0.0.1 --master-port 29533 ./advance_mu_t_slab_driver_f64 4096 60 4096 20
!   (or mpirun with the same variables exported).  Without them: one rank, no communicator.
!   AMT_RENDEZVOUS_FILE overrides the file through which rank 0 hands out the communicator id.
!   AMT_SLAB_REFRESH=1: one amt_slab_step per sweep with new values in the exchanged fields and re-poisoned halo rows before each
!   (see next_inputs below); the timed figure then includes nothing but the sweeps (the refills are outside the events).
!
!   advance_mu_t_slab_driver [NI NK NJ [nsweeps [loopback]]]
!     loopback = 1: one-rank self test, the rank is its own neighbour (exercises RCCL and the
!     two-stream schedule on a single GPU)
program advance_mu_t_slab_driver
  use iso_c_binding
  use amt_c_binding
  implicit none

  integer, parameter :: wp = kind(1.0)          ! default REAL: fp32, or fp64 with -fdefault-real-8
  integer :: ni, nk, nj, nsweeps, loopback, rank, world, local_rank, nrows, jlo, jhi
  integer :: ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte
  integer :: idim, flags
  character(len=256) :: arg
  character(len=512) :: path, port
  character(kind=c_char), target :: uid(128)
  character(kind=c_char), allocatable :: cpath(:)
  integer(c_int64_t), parameter :: seed = 12345_c_int64_t
  type(c_ptr) :: dom, slab, idptr
  real(c_float) :: ms, ms1
  integer :: sides
  logical :: refresh
  real(wp), allocatable, target :: mu(:,:)
  real(kind=8) :: cells, bytes
  real(c_double) :: ms_job
  integer(c_int) :: crank, cworld
  integer(kind=8) :: c0, c1, crate
  integer :: n

  ni = 512; nk = 60; nj = 512; nsweeps = 20; loopback = 0
  if (command_argument_count() >= 3) then
     call get_command_argument(1, arg); read (arg, *) ni
     call get_command_argument(2, arg); read (arg, *) nk
     call get_command_argument(3, arg); read (arg, *) nj
  end if
  if (command_argument_count() >= 4) then
     call get_command_argument(4, arg); read (arg, *) nsweeps
  end if
  if (command_argument_count() >= 5) then
     call get_command_argument(5, arg); read (arg, *) loopback
  end if
  rank = env_int('RANK', 0)
  world = env_int('WORLD_SIZE', 1)
  local_rank = env_int('LOCAL_RANK', rank)
  if (nj < world) error stop 'fewer rows than ranks'
  call amt_check(amt_set_device(int(local_rank, c_int)), 'amt_set_device')

  ! the domain (SURVEY.md section 8 convention), i memory padded to whole 32-element runs ...
  ids = 1; ide = ni + 1; jds = 1; jde = nj + 1; kde = nk + 1
  ims = 1 - 32
  idim = ((ni + 1 - ims + 1 + 31) / 32) * 32
  ime = ims + idim - 1
  kms = 1; kme = nk + 1; kts = 1; kte = kde; its = 1; ite = ide
  ! ... and this rank's rows of it: one halo row each side in memory
  nrows = jde - jds
  jlo = jds + (nrows * rank) / world
  jhi = jds + (nrows * (rank + 1)) / world - 1
  jms = jlo - 1; jme = jhi + 1; jts = jlo; jte = jhi

  call amt_check(amt_domain_create(dom, int(storage_size(1.0_wp)/8, c_int), 0_c_int, 0_c_int, 0_c_int,  &
                                   ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,            &
                                   its, ite, jts, jte, kts, kte), 'amt_domain_create')
  call amt_check(amt_domain_fill_synthetic(dom, seed, int(ims, c_long), int(kms - 1, c_long), int(jms, c_long), &
                                           int(ni + 2, c_long), int(nk + 1, c_long), int(nj + 2, c_long)), &
                 'amt_domain_fill_synthetic')

  flags = 0
  idptr = c_null_ptr
  if (world > 1 .or. loopback /= 0) then
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
     ! nonce 0: derived from the launcher (run id, parent process), so that a file left by an
     ! earlier launch on the same port is never taken for this one's
     call amt_check(amt_comm_rendezvous_file(cpath, 0_c_int64_t, int(rank, c_int), int(max(world, 1), c_int), &
                                             120.0_c_double, uid), 'amt_comm_rendezvous_file')
     idptr = c_loc(uid)
     if (loopback /= 0) flags = 2                      ! AMT_SLAB_LOOPBACK
  end if
  call amt_check(amt_slab_create(slab, dom, int(rank, c_int), int(world, c_int), idptr, int(flags, c_int)), 'amt_slab_create')

  refresh = env_int('AMT_SLAB_REFRESH', 0) /= 0
  sides = 0
  if (rank > 0 .or. loopback /= 0) sides = sides + AMT_SIDE_BELOW
  if (rank < world - 1 .or. loopback /= 0) sides = sides + AMT_SIDE_ABOVE
  if (refresh) call amt_check(amt_domain_poison_halos(dom, int(sides, c_int)), 'amt_domain_poison_halos')
  call next_inputs(1)
  call amt_check(amt_slab_step(slab, 1_c_int), 'amt_slab_step (warm-up)')   ! code objects, RCCL connections
  call next_inputs(2)
  call amt_check(amt_slab_step(slab, 1_c_int), 'amt_slab_step (warm-up)')
  call amt_check(amt_slab_barrier(slab), 'amt_slab_barrier')                ! every rank starts the clock together
  call system_clock(c0, crate)
  if (refresh) then
     ms = 0.0
     do n = 1, nsweeps
        call next_inputs(2 + n)
        call amt_check(amt_slab_step_timed(slab, 1_c_int, ms1), 'amt_slab_step_timed')
        ms = ms + ms1
     end do
  else
     call amt_check(amt_slab_step_timed(slab, int(nsweeps, c_int), ms), 'amt_slab_step_timed')
  end if
  call amt_check(amt_slab_sync(slab), 'amt_slab_sync')
  call system_clock(c1)
  ! the sweep time of the JOB is the slowest rank's (a rank's own event time excludes the wait
  ! for a late neighbour's halo only if that neighbour is late AFTER this rank has finished)
  ms_job = real(c1 - c0, 8) * 1.0d3 / real(crate, 8)
  call amt_check(amt_slab_max(slab, ms_job), 'amt_slab_max')
  call amt_check(amt_slab_comm_info(slab, crank, cworld), 'amt_slab_comm_info')

  cells = real(ni, 8) * real(nk, 8) * real(jhi - jlo + 1, 8)
  bytes = real(storage_size(1.0_wp)/8, 8) * real(ni, 8) * real(jhi - jlo + 1, 8) * (11.0d0 * nk + 14.0d0)
  allocate (mu(ims:ime, jms:jme))
  call amt_check(amt_domain_download(dom, AMT_F_MU, c_loc(mu)), 'amt_domain_download')
  print '(a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,f9.4,a,f11.1,a,f8.1,a,i0,a,es22.14)',                       &
        'rank ', rank, '/', world, ': rows ', jlo, '..', jhi, ' of ', ni, 'x', nk, ' real*x  ',       &
        ms / nsweeps, ' ms/sweep ', cells * nsweeps / (ms * 1.0d-3) / 1.0d6, ' Mcells/s ',            &
        bytes * nsweeps / (ms * 1.0d-3) / 1.0d9, ' GB/s algorithmic; halo bytes/sweep ',              &
        amt_slab_halo_bytes(slab), '; sum(mu) ', sum(real(mu(1:ni, jlo:jhi), 8))
  if (rank == 0) then
     cells = real(ni, 8) * real(nk, 8) * real(nj, 8)
     print '(a,i0,a,f9.4,a,f11.1,a)', 'job: ', cworld, ' rank(s) in the communicator, slowest rank ', &
           ms_job / nsweeps, ' ms/sweep wall = ', cells * nsweeps / (ms_job * 1.0d-3) / 1.0d6, ' Mcells/s'
  end if

  call amt_check(amt_slab_destroy(slab), 'amt_slab_destroy')
  call amt_check(amt_domain_destroy(dom), 'amt_domain_destroy')

contains

  ! AMT_SLAB_REFRESH=1: before sweep number `sweep` (1-based; not before the first) the fields that cross a slab boundary get new
  ! values -- amt_domain_fill_fields with seed + sweep - 1 stands in for advance_uv rewriting u, v before every advance_mu_t call --
  ! and the halo rows are poisoned with NaN again: only an exchange that delivers every sweep then gives the unsplit run's result
  subroutine next_inputs(sweep)
    integer, intent(in) :: sweep
    if (.not. refresh .or. sweep <= 1) return
    call amt_check(amt_domain_fill_fields(dom, AMT_EXCHANGED_FIELDS, seed + int(sweep - 1, c_int64_t), int(ims, c_long),     &
                                          int(kms - 1, c_long), int(jms, c_long), int(ni + 2, c_long), int(nk + 1, c_long), &
                                          int(nj + 2, c_long)), 'amt_domain_fill_fields')
    call amt_check(amt_domain_poison_halos(dom, int(sides, c_int)), 'amt_domain_poison_halos')
  end subroutine next_inputs

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

end program advance_mu_t_slab_driver
