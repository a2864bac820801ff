! oracle/ref_shim.f90 -- TEST INFRASTRUCTURE ONLY.
!
! C-callable shim around the REFERENCE routine itself: it USEs the reference's
! module_small_step_em / module_configure, which oracle/Makefile compiles from
! the sources where they lie under /root/reference (nothing is copied into this
! repository; objects, .mod files and the resulting shared library go to
! oracle/_ref/, which is git-ignored).  Default REAL is used throughout, so the
! same file gives the fp32 library and, with -fdefault-real-8, the fp64 library.
!
! The reference routine dumps five arrays into the current directory on every
! call (module_small_step_em.f90:175-189); oracle/oracle.py runs it inside a
! scratch directory whose five file names are symlinks to /dev/null.
subroutine ref_advance_mu_t( ww, ww_1, u, u_1, v, v_1,            &
                             mu, mut, muave, muts, muu, muv,      &
                             mudf, t, t_1,                        &
                             t_ave, ft, mu_tend,                  &
                             rdx, rdy, dts, epssm,                &
                             dnw, fnm, fnp, rdnw,                 &
                             msfuy, msfvx_inv,                    &
                             msftx, msfty,                        &
                             periodic_x, specified, nested,       &
                             ids, ide, jds, jde, kde,             &
                             ims, ime, jms, jme, kms, kme,        &
                             its, ite, jts, jte, kts, kte ) bind(C, name="ref_advance_mu_t")
  use iso_c_binding, only : c_int
  use module_configure, only : grid_config_rec_type
  use module_small_step_em, only : advance_mu_t
  implicit none
  integer(c_int), value :: periodic_x, specified, nested
  integer(c_int), value :: ids, ide, jds, jde, kde
  integer(c_int), value :: ims, ime, jms, jme, kms, kme
  integer(c_int), value :: its, ite, jts, jte, kts, kte
  real, dimension(ims:ime, kms:kme, jms:jme) :: ww, ww_1, u, u_1, v, v_1, t, t_1, t_ave, ft
  real, dimension(ims:ime, jms:jme) :: mu, mut, muave, muts, muu, muv, mudf, mu_tend
  real, dimension(ims:ime, jms:jme) :: msfuy, msfvx_inv, msftx, msfty
  real, dimension(kms:kme) :: dnw, fnm, fnp, rdnw
  real, value :: rdx, rdy, dts, epssm
  type(grid_config_rec_type), save :: config_flags

  config_flags%periodic_x = (periodic_x /= 0)
  config_flags%specified  = (specified  /= 0)
  config_flags%nested     = (nested     /= 0)

  call advance_mu_t( ww, ww_1, u, u_1, v, v_1,            &
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
end subroutine ref_advance_mu_t
