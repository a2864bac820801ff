! oracle/fortran/advance_mu_t_cpu.f90 -- TEST INFRASTRUCTURE ONLY (CPU baseline, never the product).
!
! The build's own Fortran-90 restatement of WRF-ARW's advance_mu_t for the host cores: the
! "Fortran CPU path" of BASELINE.json's north_star and SURVEY.md section 7 step 2 / section 8(d).
! Same 48-argument signature and (ims:ime,kms:kme,jms:jme) layout as the reference routine
! (/root/reference/module_small_step_em.f90:7-78), but a different program:
!
!   * ONE pass over j instead of the reference's three (ww/mu :112-172, theta pre-update :208-215,
!     flux-form theta :217-250): a column never reads another column's outputs (SURVEY.md section 3),
!     so row j is finished -- mass, omega, theta -- before row j+1 is touched;
!   * i is blocked (AMT_IB columns) and the blocks MARCH in j (i block outer, j inner -- the order of the
!     GPU kernel): rows j and j+1 of v, v_1 and rows j-1, j, j+1 of t_1 of a block stay in the core's
!     cache from one row to the next, so every input element comes from memory once (with j outer and
!     the blocks inner, as in r03, a whole row of all blocks passed between two uses and v, v_1 were
!     fetched twice, t_1 three times: 18 words per cell against 14 now, write-allocates included);
!   * the omega recurrence (:159-163), its perturbation form (:168-172) and the vertical theta flux
!     (:224-229) are ONE k loop: ww is written once and never read back (the un-subtracted value of the
!     level below is carried in a per-column scalar, exactly the value the reference reads from ww(k-1));
!   * the k-column scratch is one block wide and belongs to the calling thread (the reference's
!     (its:ite,kts:kte) automatics are 2 x 2 MB at 4096 columns);
!   * no debug dumps (:175-189 are a side effect of the sample, 99.6 % of its wall time);
!   * OpenMP over j-tiles in the C-callable drivers below: every thread calls the routine with its own
!     jts:jte, the scheme sketched in the reference driver (advance_mu_t_driver.f90:175-209).
!
! Build knobs (-cpp): AMT_IB columns per i block (default 1024: the fastest of 64 ... 4096 on 16 cores of 2 x EPYC 9575F,
! profiles/r04_raw/cpu_fortran_tune.txt), AMT_J_OUTER=1 the r03 loop order (A/B only).
!
! Every expression keeps the reference's association and the file is compiled with
! -ffp-contract=off, so the results are the reference's bits: tests/test_fortran_cpu.py holds it
! against tests/golden/ (outputs of the compiled reference) and the C oracle.
!
! Only tests/ and bench.py's cpu_baseline leg load the library built from this file.
module advance_mu_t_cpu_mod
  use module_configure, only : grid_config_rec_type
  implicit none
  private
  public :: advance_mu_t_cpu
#ifndef AMT_IB
#define AMT_IB 1024
#endif
  integer, parameter :: IB = AMT_IB       ! columns per i block (scratch: 2 x IB x kde reals)

contains

  subroutine advance_mu_t_cpu( ww, ww_1, u, u_1, v, v_1,            &
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
    type(grid_config_rec_type), intent(in) :: config_flags
    integer, intent(in) :: ids, ide, jds, jde, kde
    integer, intent(in) :: ims, ime, jms, jme, kms, kme
    integer, intent(in) :: its, ite, jts, jte, kts, kte
    real, dimension(ims:ime, kms:kme, jms:jme), intent(in)    :: u, v, u_1, v_1, t_1, ft, ww_1
    real, dimension(ims:ime, kms:kme, jms:jme), intent(inout) :: ww, t, t_ave
    real, dimension(ims:ime, jms:jme), intent(in)    :: muu, muv, mut, msfuy, msfvx_inv, msftx, msfty, mu_tend
    real, dimension(ims:ime, jms:jme), intent(out)   :: muave, muts, mudf
    real, dimension(ims:ime, jms:jme), intent(inout) :: mu
    real, dimension(kms:kme), intent(in) :: fnm, fnp, dnw, rdnw
    real, intent(in) :: rdx, rdy, dts, epssm

    ! one block wide, kept by the calling thread from call to call (threadprivate: a fresh 1 MB allocation per
    ! call and thread is an mmap / page-fault / munmap round trip on every sweep)
    real, allocatable, save :: div(:, :)     ! (IB, kts:kte)   horizontal mass-flux divergence of the block's columns
    real, allocatable, save :: flx(:, :)     ! (IB, kts:kte+1) vertical theta flux at the level interfaces
    !$omp threadprivate(div, flx)
    real :: colsum(IB)            ! its column integral
    real :: wwu(IB)               ! omega of the recurrence (:161) at the current level, before :170
    real :: old_mu
    integer :: i, j, k, ib0, ib1, n, ic
    integer :: i_lo, i_hi, j_lo, j_hi, k_hi

    ! on the heap: 2 x IB x kde reals would not fit an OpenMP worker's stack at a few hundred levels
    if (allocated(div)) then
      if (lbound(div, 2) /= kts .or. ubound(div, 2) /= kte) deallocate(div, flx)
    end if
    if (.not. allocated(div)) allocate(div(IB, kts:kte), flx(IB, kts:kte+1))

    ! the compute window (reference :91-106)
    i_lo = its
    i_hi = min(ite, ide - 1)
    j_lo = jts
    j_hi = min(jte, jde - 1)
    k_hi = kte - 1
    if (.not. config_flags%periodic_x) then
      if (config_flags%specified .or. config_flags%nested) then
        i_lo = max(its, ids + 1)
        i_hi = min(ite, ide - 2)
      end if
    end if
    if (config_flags%specified .or. config_flags%nested) then
      j_lo = max(jts, jds + 1)
      j_hi = min(jte, jde - 2)
    end if

#if AMT_J_OUTER
    do j = j_lo, j_hi
      do ib0 = i_lo, i_hi, IB
#else
    do ib0 = i_lo, i_hi, IB
      do j = j_lo, j_hi
#endif
        ib1 = min(ib0 + IB - 1, i_hi)
        n = ib1 - ib0 + 1

        ! --- divergence and its column integral (reference :140-149) ---
        colsum(1:n) = 0.
        do k = kts, k_hi
          do ic = 1, n
            i = ib0 + ic - 1
            div(ic, k) = msftx(i, j) * msfty(i, j) * (                                              &
                 rdy * ( (v(i, k, j+1) + muv(i, j+1) * v_1(i, k, j+1) * msfvx_inv(i, j+1))          &
                       - (v(i, k, j  ) + muv(i, j  ) * v_1(i, k, j  ) * msfvx_inv(i, j  )) )        &
               + rdx * ( (u(i+1, k, j) + muu(i+1, j) * u_1(i+1, k, j) / msfuy(i+1, j))              &
                       - (u(i  , k, j) + muu(i  , j) * u_1(i  , k, j) / msfuy(i  , j)) ) )
            colsum(ic) = colsum(ic) + dnw(k) * div(ic, k)
          end do
        end do

        ! --- column mass (reference :151-157) ---
        do ic = 1, n
          i = ib0 + ic - 1
          old_mu = mu(i, j)
          mu(i, j) = mu(i, j) + dts * (colsum(ic) + mu_tend(i, j))
          mudf(i, j) = (colsum(ic) + mu_tend(i, j))
          muts(i, j) = mut(i, j) + mu(i, j)
          muave(i, j) = .5 * ((1. + epssm) * mu(i, j) + (1. - epssm) * old_mu)
        end do

        ! --- omega: upward recurrence from the incoming ww(:,1,:) (reference :159-163), the perturbation
        !     form (:168-172; it runs after the whole recurrence, so the recurrence sees the un-subtracted
        !     value below: carried in wwu), and the vertical theta flux at the interfaces (:219-229) ---
        flx(1:n, 1) = 0.
        flx(1:n, kde) = 0.
        do ic = 1, n
          i = ib0 + ic - 1
          wwu(ic) = ww(i, 1, j)
          ww(i, 1, j) = wwu(ic) - ww_1(i, 1, j)
        end do
        do k = 2, k_hi
          do ic = 1, n
            i = ib0 + ic - 1
            wwu(ic) = wwu(ic) - dnw(k-1) * (colsum(ic) + div(ic, k-1) + mu_tend(i, j)) / msfty(i, j)
            ww(i, k, j) = wwu(ic) - ww_1(i, k, j)
            flx(ic, k) = ww(i, k, j) * (fnm(k) * t_1(i, k, j) + fnp(k) * t_1(i, k-1, j))
          end do
        end do

        ! --- theta: save, tendency, flux-form advection (reference :211-212 and :234-248) ---
        do k = 1, k_hi
          do ic = 1, n
            i = ib0 + ic - 1
            t_ave(i, k, j) = t(i, k, j)
            t(i, k, j) = t(i, k, j) + msfty(i, j) * dts * ft(i, k, j)
            t(i, k, j) = t(i, k, j) - dts * msfty(i, j) * (                                         &
                 msftx(i, j) * (                                                                    &
                   .5 * rdy * ( v(i, k, j+1) * (t_1(i, k, j+1) + t_1(i, k, j  ))                    &
                              - v(i, k, j  ) * (t_1(i, k, j  ) + t_1(i, k, j-1)) )                  &
                 + .5 * rdx * ( u(i+1, k, j) * (t_1(i+1, k, j) + t_1(i  , k, j))                    &
                              - u(i  , k, j) * (t_1(i  , k, j) + t_1(i-1, k, j)) ) )                &
               + rdnw(k) * (flx(ic, k+1) - flx(ic, k)) )
          end do
        end do
      end do
    end do
  end subroutine advance_mu_t_cpu

end module advance_mu_t_cpu_mod

! C-callable driver, one per precision build (default REAL; -fdefault-real-8 gives the fp64 library):
!   int amt_cpu_fortran(ww, ..., msfty, periodic_x, specified, nested, ids, ..., kte, nthreads)
! the argument list of oracle_advance_mu_t_f32/_f64 plus the number of OpenMP j-tiles.
function amt_cpu_fortran( ww, ww_1, u, u_1, v, v_1,            &
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
                          its, ite, jts, jte, kts, kte, nthreads ) result(rc) bind(C, name="amt_cpu_fortran")
  use iso_c_binding, only : c_int
  use module_configure, only : grid_config_rec_type
  use advance_mu_t_cpu_mod, only : advance_mu_t_cpu
  implicit none
  integer(c_int), value :: periodic_x, specified, nested
  integer(c_int), value :: ids, ide, jds, jde, kde
  integer(c_int), value :: ims, ime, jms, jme, kms, kme
  integer(c_int), value :: its, ite, jts, jte, kts, kte, nthreads
  real, dimension(ims:ime, kms:kme, jms:jme) :: ww, ww_1, u, u_1, v, v_1, t, t_1, t_ave, ft
  real, dimension(ims:ime, jms:jme) :: mu, mut, muave, muts, muu, muv, mudf, mu_tend
  real, dimension(ims:ime, jms:jme) :: msfuy, msfvx_inv, msftx, msfty
  real, dimension(kms:kme) :: dnw, fnm, fnp, rdnw
  real, value :: rdx, rdy, dts, epssm
  integer(c_int) :: rc
  type(grid_config_rec_type) :: config_flags
  integer :: id, nt, nrow, lo, hi

  rc = 0
  ! what the Fortran defines: literal levels 1, 2 and flx(:,kde) (reference :159,:168,:221)
  if (kts /= 1 .or. kte /= kde) then
    rc = 2
    return
  end if
  config_flags%periodic_x = (periodic_x /= 0)
  config_flags%specified  = (specified  /= 0)
  config_flags%nested     = (nested     /= 0)
  nrow = jte - jts + 1
  nt = max(1, min(int(nthreads), nrow))
  !$omp parallel do num_threads(nt) schedule(static, 1) private(lo, hi)
  do id = 0, nt - 1
    lo = jts + int((int(nrow, 8) * id) / nt)
    hi = jts + int((int(nrow, 8) * (id + 1)) / nt) - 1
    if (hi >= lo) then
      call advance_mu_t_cpu( ww, ww_1, u, u_1, v, v_1, mu, mut, muave, muts, muu, muv,    &
                             mudf, t, t_1, t_ave, ft, mu_tend, rdx, rdy, dts, epssm,      &
                             dnw, fnm, fnp, rdnw, msfuy, msfvx_inv, msftx, msfty,         &
                             config_flags, ids, ide, jds, jde, kde,                       &
                             ims, ime, jms, jme, kms, kme, its, ite, lo, hi, kts, kte )
    end if
  end do
  !$omp end parallel do
end function amt_cpu_fortran
