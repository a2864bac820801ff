! module_small_step_em.f90 -- drop-in for the reference module of the same name.
!
! SUBROUTINE advance_mu_t has the identical 48-argument signature, argument
! intents and (ims:ime, kms:kme, jms:jme) array layout as the reference
! (module_small_step_em.f90:7-78), so a caller (solve_em, or the reference's
! advance_mu_t_driver.f90:193-205) links against this module unchanged.  The body
! is one call through ISO_C_BINDING into the MI355X library
! (include/amt_advance_mu_t.h, amt_advance_mu_t_f32 / _f64 chosen by the kind of
! default REAL, i.e. by -fdefault-real-8): host arrays in, host arrays out.
! There is no Fortran compute path here; without a GPU the call stops with the
! library's error text.
MODULE module_small_step_em

USE module_configure, ONLY : grid_config_rec_type

CONTAINS

SUBROUTINE advance_mu_t( ww, ww_1, u, u_1, v, v_1,            &
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
                         its, ite, jts, jte, kts, kte        )

  USE iso_c_binding
  USE amt_c_binding
  IMPLICIT NONE

  TYPE(grid_config_rec_type), INTENT(IN   ) :: config_flags

  INTEGER,      INTENT(IN   )    :: ids,ide, jds,jde, kde
  INTEGER,      INTENT(IN   )    :: ims,ime, jms,jme, kms,kme
  INTEGER,      INTENT(IN   )    :: its,ite, jts,jte, kts,kte

  REAL, DIMENSION( ims:ime , kms:kme, jms:jme ), INTENT(IN   ), TARGET :: u, v, u_1, v_1, t_1, ft
  REAL, DIMENSION( ims:ime , kms:kme, jms:jme ), INTENT(INOUT), TARGET :: ww, ww_1, t, t_ave
  REAL, DIMENSION( ims:ime , jms:jme ), INTENT(IN   ), TARGET :: muu, muv, mut, msfuy, msfvx_inv, &
                                                                 msftx, msfty, mu_tend
  REAL, DIMENSION( ims:ime , jms:jme ), INTENT(  OUT), TARGET :: muave, muts, mudf
  REAL, DIMENSION( ims:ime , jms:jme ), INTENT(INOUT), TARGET :: mu
  REAL, DIMENSION( kms:kme ),           INTENT(IN   ), TARGET :: fnm, fnp, dnw, rdnw
  REAL,                                 INTENT(IN   ) :: rdx, rdy, dts, epssm

  INTEGER(c_int) :: rc, px, sp, ne

  px = merge(1_c_int, 0_c_int, config_flags%periodic_x)
  sp = merge(1_c_int, 0_c_int, config_flags%specified)
  ne = merge(1_c_int, 0_c_int, config_flags%nested)

  IF ( kind(rdx) == c_double ) THEN
     rc = amt_advance_mu_t_f64( c_loc(ww), c_loc(ww_1), c_loc(u), c_loc(u_1), c_loc(v), c_loc(v_1),     &
                                c_loc(mu), c_loc(mut), c_loc(muave), c_loc(muts), c_loc(muu), c_loc(muv), &
                                c_loc(mudf), c_loc(t), c_loc(t_1), c_loc(t_ave), c_loc(ft), c_loc(mu_tend), &
                                real(rdx, c_double), real(rdy, c_double), real(dts, c_double),          &
                                real(epssm, c_double),                                                  &
                                c_loc(dnw), c_loc(fnm), c_loc(fnp), c_loc(rdnw),                        &
                                c_loc(msfuy), c_loc(msfvx_inv), c_loc(msftx), c_loc(msfty),             &
                                px, sp, ne,                                                             &
                                ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,                  &
                                its, ite, jts, jte, kts, kte )
  ELSE
     rc = amt_advance_mu_t_f32( c_loc(ww), c_loc(ww_1), c_loc(u), c_loc(u_1), c_loc(v), c_loc(v_1),     &
                                c_loc(mu), c_loc(mut), c_loc(muave), c_loc(muts), c_loc(muu), c_loc(muv), &
                                c_loc(mudf), c_loc(t), c_loc(t_1), c_loc(t_ave), c_loc(ft), c_loc(mu_tend), &
                                real(rdx, c_float), real(rdy, c_float), real(dts, c_float),             &
                                real(epssm, c_float),                                                   &
                                c_loc(dnw), c_loc(fnm), c_loc(fnp), c_loc(rdnw),                        &
                                c_loc(msfuy), c_loc(msfvx_inv), c_loc(msftx), c_loc(msfty),             &
                                px, sp, ne,                                                             &
                                ids, ide, jds, jde, kde, ims, ime, jms, jme, kms, kme,                  &
                                its, ite, jts, jte, kts, kte )
  END IF
  CALL amt_check(rc, 'advance_mu_t')

END SUBROUTINE advance_mu_t

END MODULE module_small_step_em
