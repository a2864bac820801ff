! module_configure.f90 -- stand-in for WRF's Registry-generated configuration type.
!
! The real grid_config_rec_type has 1 796 scalar fields (reference
! module_configure.f90:3-1800); advance_mu_t reads three logicals of it
! (module_small_step_em.f90:97-103).  A host that already has WRF's own
! module_configure uses that one instead of this file -- the wrapper in
! module_small_step_em.f90 only touches these three components.
MODULE module_configure

   TYPE grid_config_rec_type
      logical :: specified  = .false.
      logical :: periodic_x = .false.
      logical :: nested     = .false.
   END TYPE grid_config_rec_type

END MODULE module_configure
