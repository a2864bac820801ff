! amt_c_binding.f90 -- ISO_C_BINDING interfaces of include/amt_advance_mu_t.h.
!
! Arrays cross the boundary as C addresses (c_loc of the first element): the
! C-ABI takes plain pointers and the Fortran-style inclusive bounds unchanged.
MODULE amt_c_binding
   use iso_c_binding
   implicit none

   integer(c_int), parameter :: AMT_OK = 0
   ! enum amt_slab_flags (amt_slab_create / amt_grid_create)
   integer(c_int), parameter :: AMT_SLAB_NO_OVERLAP = 1, AMT_SLAB_LOOPBACK = 2, AMT_SLAB_TRANSPORT_IPC = 4

   interface
      ! (1) one-shot host drop-ins
      function amt_advance_mu_t_f32(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv,    &
                                    mudf, t, t_1, t_ave, ft, mu_tend, rdx, rdy, dts, epssm,      &
                                    dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty,         &
                                    periodic_x, specified, nested,                               &
                                    ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,       &
                                    its, ite, jts, jte, kts, kte) bind(C, name="amt_advance_mu_t_f32") result(rc)
         import :: c_ptr, c_float, c_int
         type(c_ptr), value :: ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv
         type(c_ptr), value :: mudf, t, t_1, t_ave, ft, mu_tend
         real(c_float), value :: rdx, rdy, dts, epssm
         type(c_ptr), value :: dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty
         integer(c_int), value :: periodic_x, specified, nested
         integer(c_int), value :: ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme
         integer(c_int), value :: its, ite, jts, jte, kts, kte
         integer(c_int) :: rc
      end function
      function amt_advance_mu_t_f64(ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv,    &
                                    mudf, t, t_1, t_ave, ft, mu_tend, rdx, rdy, dts, epssm,      &
                                    dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty,         &
                                    periodic_x, specified, nested,                               &
                                    ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,       &
                                    its, ite, jts, jte, kts, kte) bind(C, name="amt_advance_mu_t_f64") result(rc)
         import :: c_ptr, c_double, c_int
         type(c_ptr), value :: ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv
         type(c_ptr), value :: mudf, t, t_1, t_ave, ft, mu_tend
         real(c_double), value :: rdx, rdy, dts, epssm
         type(c_ptr), value :: dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty
         integer(c_int), value :: periodic_x, specified, nested
         integer(c_int), value :: ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme
         integer(c_int), value :: its, ite, jts, jte, kts, kte
         integer(c_int) :: rc
      end function

      ! page-lock / release a host array so that the one-shot call streams it
      function amt_host_pin(ptr, bytes) bind(C, name="amt_host_pin") result(rc)
         import :: c_ptr, c_size_t, c_int
         type(c_ptr), value :: ptr
         integer(c_size_t), value :: bytes
         integer(c_int) :: rc
      end function
      function amt_host_unpin(ptr) bind(C, name="amt_host_unpin") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: ptr
         integer(c_int) :: rc
      end function
      ! one call, several devices: the one-shot calls of this thread fan their tile's rows jts..jte over n device slots (ids may
      ! repeat), halo rows from the host arrays -- what the reference's own host call does (advance_mu_t_no_async.cu:108-162);
      ! n = 0 turns it off.  AMT_ONESHOT_DEVICES="0,1,2" | "all" in the environment does the same without a call.
      function amt_host_set_devices(n, device_ids) bind(C, name="amt_host_set_devices") result(rc)
         import :: c_int
         integer(c_int), value :: n
         integer(c_int), intent(in) :: device_ids(*)
         integer(c_int) :: rc
      end function
      function amt_host_devices(device_ids, cap) bind(C, name="amt_host_devices") result(n)
         import :: c_int
         integer(c_int), intent(out) :: device_ids(*)
         integer(c_int), value :: cap
         integer(c_int) :: n
      end function
      ! free the device workspace the one-shot calls of this thread keep between calls
      function amt_host_release() bind(C, name="amt_host_release") result(rc)
         import :: c_int
         integer(c_int) :: rc
      end function
      ! residency cache of the one-shot calls of this thread: ww_1, u_1, v_1, t_1, ft stay on the device
      ! between calls (the sub-steps of one Runge-Kutta stage) and go up again only after amt_host_invalidate
      ! (c_null_ptr: all of them -- a new stage); amt_host_cache_check(1): checksum debug mode
      function amt_host_cache_enable(on) bind(C, name="amt_host_cache_enable") result(rc)
         import :: c_int
         integer(c_int), value :: on
         integer(c_int) :: rc
      end function
      function amt_host_cache_check(on) bind(C, name="amt_host_cache_check") result(rc)
         import :: c_int
         integer(c_int), value :: on
         integer(c_int) :: rc
      end function
      function amt_host_invalidate(ptr) bind(C, name="amt_host_invalidate") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: ptr
         integer(c_int) :: rc
      end function
      ! deferred outputs of the one-shot calls of this thread: ww, t, t_ave, mu, muave, muts, mudf (c_null_ptr: all) stay
      ! on the device after a call until amt_host_fetch brings the window's cells down; amt_host_stale: 1 while the device
      ! copy is newer than the host array
      function amt_host_defer(ptr, on) bind(C, name="amt_host_defer") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: ptr
         integer(c_int), value :: on
         integer(c_int) :: rc
      end function
      function amt_host_fetch(ptr) bind(C, name="amt_host_fetch") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: ptr
         integer(c_int) :: rc
      end function
      function amt_host_stale(ptr) bind(C, name="amt_host_stale") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: ptr
         integer(c_int) :: rc
      end function

      ! error text of the calling thread (NUL-terminated C string)
      function amt_last_error() bind(C, name="amt_last_error") result(msg)
         import :: c_ptr
         type(c_ptr) :: msg
      end function
      function amt_device_count() bind(C, name="amt_device_count") result(n)
         import :: c_int
         integer(c_int) :: n
      end function

      ! (3) resident domain handle
      function amt_domain_create(handle, dtype_bytes, periodic_x, specified, nested,             &
                                 ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,          &
                                 its, ite, jts, jte, kts, kte) bind(C, name="amt_domain_create") result(rc)
         import :: c_ptr, c_int
         type(c_ptr) :: handle                      ! amt_domain **
         integer(c_int), value :: dtype_bytes, periodic_x, specified, nested
         integer(c_int), value :: ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme
         integer(c_int), value :: its, ite, jts, jte, kts, kte
         integer(c_int) :: rc
      end function
      function amt_domain_destroy(handle) bind(C, name="amt_domain_destroy") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: handle
         integer(c_int) :: rc
      end function
      function amt_domain_set_scalars(handle, rdx, rdy, dts, epssm) bind(C, name="amt_domain_set_scalars") result(rc)
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: handle
         real(c_double), value :: rdx, rdy, dts, epssm
         integer(c_int) :: rc
      end function
      function amt_domain_upload(handle, field, host) bind(C, name="amt_domain_upload") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: handle, host
         integer(c_int), value :: field
         integer(c_int) :: rc
      end function
      function amt_domain_download(handle, field, host) bind(C, name="amt_domain_download") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: handle, host
         integer(c_int), value :: field
         integer(c_int) :: rc
      end function
      function amt_domain_step(handle, n_sweeps) bind(C, name="amt_domain_step") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: handle
         integer(c_int), value :: n_sweeps
         integer(c_int) :: rc
      end function
      ! placement tuning: the handle's arrays allocated `tries` times, two sweeps timed on each set, the fastest kept;
      ! contents unchanged; ms_per_try receives the sweep time of every set
      function amt_domain_tune_placement(handle, tries, ms_per_try) bind(C, name="amt_domain_tune_placement") result(rc)
         import :: c_ptr, c_int, c_float
         type(c_ptr), value :: handle
         integer(c_int), value :: tries
         real(c_float) :: ms_per_try(*)
         integer(c_int) :: rc
      end function
      function amt_domain_step_timed(handle, n_sweeps, ms_total) bind(C, name="amt_domain_step_timed") result(rc)
         import :: c_ptr, c_int, c_float
         type(c_ptr), value :: handle
         integer(c_int), value :: n_sweeps
         real(c_float) :: ms_total
         integer(c_int) :: rc
      end function
      function amt_domain_sync(handle) bind(C, name="amt_domain_sync") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: handle
         integer(c_int) :: rc
      end function

      function amt_domain_fill_synthetic(handle, seed, gi0, gk0, gj0, gidim, gkdim, gjdim)        &
            bind(C, name="amt_domain_fill_synthetic") result(rc)
         import :: c_ptr, c_int, c_int64_t, c_long
         type(c_ptr), value :: handle
         integer(c_int64_t), value :: seed
         integer(c_long), value :: gi0, gk0, gj0, gidim, gkdim, gjdim
         integer(c_int) :: rc
      end function

      ! the fields of field_mask only (bit f = field id f; AMT_EXCHANGED_FIELDS below): the stand-in for advance_uv
      ! rewriting u, v before every advance_mu_t call
      function amt_domain_fill_fields(handle, field_mask, seed, gi0, gk0, gj0, gidim, gkdim, gjdim) &
            bind(C, name="amt_domain_fill_fields") result(rc)
         import :: c_ptr, c_int, c_int64_t, c_long
         type(c_ptr), value :: handle
         integer(c_int64_t), value :: field_mask, seed
         integer(c_long), value :: gi0, gk0, gj0, gidim, gkdim, gjdim
         integer(c_int) :: rc
      end function

      ! NaN into what the stencil reads from a neighbour: sides = sum of AMT_SIDE_BELOW / ABOVE / LEFT / RIGHT
      function amt_domain_poison_halos(handle, sides) bind(C, name="amt_domain_poison_halos") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: handle
         integer(c_int), value :: sides
         integer(c_int) :: rc
      end function

      ! ---- j-slabs over several GPUs, one process per GPU (RCCL halos) ----
      function amt_set_device(device) bind(C, name="amt_set_device") result(rc)
         import :: c_int
         integer(c_int), value :: device
         integer(c_int) :: rc
      end function
      function amt_comm_rendezvous_file(path, nonce, rank, world, timeout_s, id_out)              &
            bind(C, name="amt_comm_rendezvous_file") result(rc)
         import :: c_char, c_int, c_double, c_int64_t
         character(kind=c_char), intent(in) :: path(*)      ! NUL-terminated
         integer(c_int64_t), value :: nonce                 ! 0: amt_comm_launch_nonce()
         integer(c_int), value :: rank, world
         real(c_double), value :: timeout_s
         character(kind=c_char), intent(out) :: id_out(128)
         integer(c_int) :: rc
      end function
      function amt_slab_create(slab, domain, rank, world, unique_id, flags) bind(C, name="amt_slab_create") result(rc)
         import :: c_ptr, c_int
         type(c_ptr) :: slab                         ! amt_slab **
         type(c_ptr), value :: domain, unique_id     ! unique_id may be c_null_ptr when world == 1
         integer(c_int), value :: rank, world, flags
         integer(c_int) :: rc
      end function
      function amt_slab_destroy(slab) bind(C, name="amt_slab_destroy") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: slab
         integer(c_int) :: rc
      end function
      function amt_slab_step(slab, n_sweeps) bind(C, name="amt_slab_step") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: slab
         integer(c_int), value :: n_sweeps
         integer(c_int) :: rc
      end function
      function amt_slab_step_timed(slab, n_sweeps, ms_total) bind(C, name="amt_slab_step_timed") result(rc)
         import :: c_ptr, c_int, c_float
         type(c_ptr), value :: slab
         integer(c_int), value :: n_sweeps
         real(c_float) :: ms_total
         integer(c_int) :: rc
      end function
      function amt_slab_sync(slab) bind(C, name="amt_slab_sync") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: slab
         integer(c_int) :: rc
      end function
      function amt_slab_halo_bytes(slab) bind(C, name="amt_slab_halo_bytes") result(n)
         import :: c_ptr, c_long
         type(c_ptr), value :: slab
         integer(c_long) :: n
      end function
      function amt_slab_comm_info(slab, rank, world) bind(C, name="amt_slab_comm_info") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: slab
         integer(c_int) :: rank, world
         integer(c_int) :: rc
      end function
      function amt_slab_barrier(slab) bind(C, name="amt_slab_barrier") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: slab
         integer(c_int) :: rc
      end function
      function amt_slab_max(slab, x) bind(C, name="amt_slab_max") result(rc)
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: slab
         real(c_double) :: x                                 ! in: this rank's value, out: the maximum
         integer(c_int) :: rc
      end function

      ! (5b) patches in i and j: rank = rj * pi + ri of pi x pj (amt_slab_* is the pi = 1 case)
      function amt_grid_create(grid, domain, ri, rj, pi, pj, unique_id, flags) bind(C, name="amt_grid_create") result(rc)
         import :: c_ptr, c_int
         type(c_ptr) :: grid                         ! amt_grid **
         type(c_ptr), value :: domain, unique_id     ! unique_id may be c_null_ptr when pi * pj == 1
         integer(c_int), value :: ri, rj, pi, pj, flags
         integer(c_int) :: rc
      end function
      function amt_grid_destroy(grid) bind(C, name="amt_grid_destroy") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: grid
         integer(c_int) :: rc
      end function
      function amt_grid_exchange(grid) bind(C, name="amt_grid_exchange") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: grid
         integer(c_int) :: rc
      end function
      function amt_grid_step(grid, n_sweeps) bind(C, name="amt_grid_step") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: grid
         integer(c_int), value :: n_sweeps
         integer(c_int) :: rc
      end function
      function amt_grid_step_timed(grid, n_sweeps, ms_total) bind(C, name="amt_grid_step_timed") result(rc)
         import :: c_ptr, c_int, c_float
         type(c_ptr), value :: grid
         integer(c_int), value :: n_sweeps
         real(c_float) :: ms_total
         integer(c_int) :: rc
      end function
      function amt_grid_sync(grid) bind(C, name="amt_grid_sync") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: grid
         integer(c_int) :: rc
      end function
      function amt_grid_halo_bytes(grid) bind(C, name="amt_grid_halo_bytes") result(n)
         import :: c_ptr, c_long
         type(c_ptr), value :: grid
         integer(c_long) :: n
      end function
      function amt_grid_comm_info(grid, rank, world) bind(C, name="amt_grid_comm_info") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: grid
         integer(c_int) :: rank, world
         integer(c_int) :: rc
      end function
      function amt_grid_barrier(grid) bind(C, name="amt_grid_barrier") result(rc)
         import :: c_ptr, c_int
         type(c_ptr), value :: grid
         integer(c_int) :: rc
      end function
      function amt_grid_max(grid, x) bind(C, name="amt_grid_max") result(rc)
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: grid
         real(c_double) :: x                                 ! in: this rank's value, out: the maximum
         integer(c_int) :: rc
      end function

      ! (4) synthetic inputs
      function amt_synth_fill_host(field, dtype_bytes, dst, seed, idim, kdim, jdim, gi0, gk0, gj0, &
                                   gidim, gkdim, gjdim) bind(C, name="amt_synth_fill_host") result(rc)
         import :: c_ptr, c_int, c_long, c_int64_t
         integer(c_int), value :: field, dtype_bytes
         type(c_ptr), value :: dst
         integer(c_int64_t), value :: seed
         integer(c_long), value :: idim, kdim, jdim, gi0, gk0, gj0, gidim, gkdim, gjdim
         integer(c_int) :: rc
      end function
   end interface

   ! enum amt_field (include/amt_synth.h): the Fortran argument order
   integer(c_int), parameter :: AMT_F_WW = 0, AMT_F_WW_1 = 1, AMT_F_U = 2, AMT_F_U_1 = 3, AMT_F_V = 4,      &
      AMT_F_V_1 = 5, AMT_F_MU = 6, AMT_F_MUT = 7, AMT_F_MUAVE = 8, AMT_F_MUTS = 9, AMT_F_MUU = 10,          &
      AMT_F_MUV = 11, AMT_F_MUDF = 12, AMT_F_T = 13, AMT_F_T_1 = 14, AMT_F_T_AVE = 15, AMT_F_FT = 16,       &
      AMT_F_MU_TEND = 17, AMT_F_DNW = 18, AMT_F_FNM = 19, AMT_F_FNP = 20, AMT_F_RDNW = 21,                  &
      AMT_F_MSFUY = 22, AMT_F_MSFVX_INV = 23, AMT_F_MSFTX = 24, AMT_F_MSFTY = 25
   ! AMT_EXCHANGED_FIELDS: u, u_1, v, v_1, t_1, muu, muv, msfuy, msfvx_inv -- what crosses a patch boundary
   integer(c_int64_t), parameter :: AMT_EXCHANGED_FIELDS = ior(ior(ior(ishft(1_c_int64_t, 2), ishft(1_c_int64_t, 3)),       &
      ior(ishft(1_c_int64_t, 4), ishft(1_c_int64_t, 5))), ior(ior(ishft(1_c_int64_t, 14), ishft(1_c_int64_t, 10)),          &
      ior(ishft(1_c_int64_t, 11), ior(ishft(1_c_int64_t, 22), ishft(1_c_int64_t, 23)))))
   ! enum amt_sides (amt_domain_poison_halos)
   integer(c_int), parameter :: AMT_SIDE_BELOW = 1, AMT_SIDE_ABOVE = 2, AMT_SIDE_LEFT = 4, AMT_SIDE_RIGHT = 8

CONTAINS

   subroutine amt_check(rc, what)
      integer(c_int), intent(in) :: rc
      character(len=*), intent(in) :: what
      character(kind=c_char), pointer :: cmsg(:)
      character(len=512) :: msg
      integer :: n
      if (rc == AMT_OK) return
      call c_f_pointer(amt_last_error(), cmsg, [512])
      msg = ' '
      do n = 1, 512
         if (cmsg(n) == c_null_char) exit
         msg(n:n) = cmsg(n)
      end do
      write (*, '(a,a,a,i0,a,a)') 'amt: ', what, ' failed with status ', rc, ': ', trim(msg)
      error stop 1
   end subroutine amt_check

END MODULE amt_c_binding
