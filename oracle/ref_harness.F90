! ref_harness.F90 - C-callable harness around the REFERENCE radsurf().
!
! TEST INFRASTRUCTURE ONLY.  This file is this repository's own code; it is
! compiled together with the reference's Fortran sources (taken where they lie
! under /root/reference, never copied) by oracle/Makefile into
! oracle/_ref/libspartacus_ref.so.  It gives the real reference the same flat
! C interface as the HIP library (include/spartacus_surface_hip.h) so that
! tests can run both on identical inputs, golden vectors can be generated, and
! bench.py can time the reference's CPU path ("cpu_baseline.kind" =
! "reference").  Nothing in the product path links or calls it.
!
! The same source is compiled a second time against THIS repository's drop-in
! module radsurf_interface (spartacus_surface_amd/fortran/radsurf_interface_hip.F90)
! with -DHARNESS_ENTRY='spartacus_radsurf_shim' into
! oracle/_ref/libspartacus_shim.so: that is the test of the Fortran boundary
! (reference derived types -> ISO_C_BINDING shim -> HIP library).
!
! It drives radsurf() in the pattern of the reference's own driver: blocks of
! nblocksize columns in an OpenMP PARALLEL DO SCHEDULE(RUNTIME) loop, with
! calc_simple_spectrum_lw before radsurf in every block, repeated nrepeat
! times, wall-clock around the loop only
! (driver/spartacus_surface_driver.F90:195-269).

#ifndef HARNESS_ENTRY
#define HARNESS_ENTRY 'spartacus_radsurf_ref'
#endif

module ref_harness_types

  use, intrinsic :: iso_c_binding

  implicit none

  integer, parameter :: SS_MAX_STREAMS = 16

  type, bind(c) :: ss_legendre_gauss
    integer(c_int32_t) :: nstream
    integer(c_int32_t) :: pad_
    real(c_double) :: mu(SS_MAX_STREAMS)
    real(c_double) :: sin_ang(SS_MAX_STREAMS)
    real(c_double) :: tan_ang(SS_MAX_STREAMS)
    real(c_double) :: weight(SS_MAX_STREAMS)
    real(c_double) :: hweight(SS_MAX_STREAMS)
    real(c_double) :: vweight(SS_MAX_STREAMS)
    real(c_double) :: vadjustment
    real(c_double) :: vadjustment2
  end type ss_legendre_gauss

  type, bind(c) :: ss_config
    integer(c_int32_t) :: do_sw, do_lw, use_sw_direct_albedo
    integer(c_int32_t) :: do_vegetation, do_urban
    integer(c_int32_t) :: n_vegetation_region_forest, n_vegetation_region_urban
    integer(c_int32_t) :: nsw, nlw
    integer(c_int32_t) :: use_symmetric_vegetation_scale_forest
    integer(c_int32_t) :: use_symmetric_vegetation_scale_urban
    integer(c_int32_t) :: iverbose
    real(c_double) :: vegetation_isolation_factor_forest
    real(c_double) :: vegetation_isolation_factor_urban
    real(c_double) :: min_vegetation_fraction
    real(c_double) :: min_building_fraction
    type(ss_legendre_gauss) :: lg_sw_forest, lg_sw_urban, lg_lw_forest, lg_lw_urban
  end type ss_config

  type, bind(c) :: ss_canopy
    integer(c_int32_t) :: ncol, ntotlay
    type(c_ptr) :: nlay, istartlay, i_representation
    type(c_ptr) :: cos_sza, dz, building_fraction, veg_fraction
    type(c_ptr) :: building_scale, veg_scale, veg_ext, veg_fsd
    type(c_ptr) :: veg_contact_fraction
    type(c_ptr) :: ground_temperature, roof_temperature, wall_temperature
    type(c_ptr) :: clear_air_temperature, veg_temperature, veg_air_temperature
  end type ss_canopy

  type, bind(c) :: ss_sw_spectral
    integer(c_int32_t) :: nspec, pad_
    type(c_ptr) :: air_ext, air_ssa, veg_ssa, ground_albedo, roof_albedo
    type(c_ptr) :: wall_albedo, ground_albedo_dir, roof_albedo_dir
    type(c_ptr) :: wall_specular_frac
  end type ss_sw_spectral

  type, bind(c) :: ss_lw_spectral
    integer(c_int32_t) :: nspec, pad_
    type(c_ptr) :: air_ext, air_ssa, veg_ssa
    type(c_ptr) :: clear_air_planck, veg_planck, veg_air_planck
    type(c_ptr) :: ground_emissivity, roof_emissivity, wall_emissivity
    type(c_ptr) :: ground_emission, roof_emission, wall_emission
  end type ss_lw_spectral

  type, bind(c) :: ss_bc_out
    type(c_ptr) :: sw_albedo, sw_albedo_dir, lw_emissivity, lw_emission
  end type ss_bc_out

  type, bind(c) :: ss_flux
    integer(c_int32_t) :: nspec, ncol, ntotlay, pad_
    type(c_ptr) :: ground_dn, ground_dn_dir, ground_net
    type(c_ptr) :: top_dn, top_dn_dir, top_net
    type(c_ptr) :: roof_in, roof_in_dir, roof_net
    type(c_ptr) :: wall_in, wall_in_dir, wall_net
    type(c_ptr) :: roof_sunlit_frac, wall_sunlit_frac, ground_sunlit_frac
    type(c_ptr) :: ground_vertical_diff
    type(c_ptr) :: clear_air_abs, veg_abs, veg_air_abs, veg_abs_dir
    type(c_ptr) :: veg_sunlit_frac
    type(c_ptr) :: flux_dn_layer_top, flux_dn_dir_layer_top, flux_up_layer_top
    type(c_ptr) :: flux_dn_layer_base, flux_dn_dir_layer_base, flux_up_layer_base
  end type ss_flux

contains

  ! Copy helpers between C pointers and allocatable Fortran arrays

  subroutine get_r1(p, n, a)
    type(c_ptr), intent(in) :: p
    integer, intent(in) :: n
    real(c_double), allocatable, intent(inout) :: a(:)
    real(c_double), pointer :: f(:)
    if (allocated(a)) deallocate(a)
    if (c_associated(p)) then
      call c_f_pointer(p, f, [n])
      allocate(a(n))
      a = f
    end if
  end subroutine get_r1

  subroutine get_r2(p, m, n, a)
    type(c_ptr), intent(in) :: p
    integer, intent(in) :: m, n
    real(c_double), allocatable, intent(inout) :: a(:,:)
    real(c_double), pointer :: f(:,:)
    if (allocated(a)) deallocate(a)
    if (c_associated(p)) then
      call c_f_pointer(p, f, [m,n])
      allocate(a(m,n))
      a = f
    end if
  end subroutine get_r2

  subroutine get_i1(p, n, a)
    type(c_ptr), intent(in) :: p
    integer, intent(in) :: n
    integer(c_int32_t), allocatable, intent(inout) :: a(:)
    integer(c_int32_t), pointer :: f(:)
    if (allocated(a)) deallocate(a)
    if (c_associated(p)) then
      call c_f_pointer(p, f, [n])
      allocate(a(n))
      a = f
    end if
  end subroutine get_i1

  subroutine put_r1(a, p, n)
    real(c_double), allocatable, intent(in) :: a(:)
    type(c_ptr), intent(in) :: p
    integer, intent(in) :: n
    real(c_double), pointer :: f(:)
    if (allocated(a) .and. c_associated(p)) then
      call c_f_pointer(p, f, [n])
      f = a
    end if
  end subroutine put_r1

  subroutine put_r2(a, p, m, n)
    real(c_double), allocatable, intent(in) :: a(:,:)
    type(c_ptr), intent(in) :: p
    integer, intent(in) :: m, n
    real(c_double), pointer :: f(:,:)
    if (allocated(a) .and. c_associated(p)) then
      call c_f_pointer(p, f, [m,n])
      f = a
    end if
  end subroutine put_r2

end module ref_harness_types


module ref_harness

  use, intrinsic :: iso_c_binding
  use ref_harness_types

  implicit none

contains

  subroutine fill_config(cc, config)
    use radsurf_config, only : config_type
    type(ss_config), intent(in) :: cc
    type(config_type), intent(inout) :: config
    config%do_sw = (cc%do_sw /= 0)
    config%do_lw = (cc%do_lw /= 0)
    config%use_sw_direct_albedo = (cc%use_sw_direct_albedo /= 0)
    config%do_vegetation = (cc%do_vegetation /= 0)
    config%do_urban = (cc%do_urban /= 0)
    config%n_vegetation_region_forest = cc%n_vegetation_region_forest
    config%n_vegetation_region_urban  = cc%n_vegetation_region_urban
    config%nsw = cc%nsw
    config%nlw = cc%nlw
    config%n_stream_sw_forest = cc%lg_sw_forest%nstream
    config%n_stream_sw_urban  = cc%lg_sw_urban%nstream
    config%n_stream_lw_forest = cc%lg_lw_forest%nstream
    config%n_stream_lw_urban  = cc%lg_lw_urban%nstream
    config%use_symmetric_vegetation_scale_forest = (cc%use_symmetric_vegetation_scale_forest /= 0)
    config%use_symmetric_vegetation_scale_urban  = (cc%use_symmetric_vegetation_scale_urban /= 0)
    config%vegetation_isolation_factor_forest = cc%vegetation_isolation_factor_forest
    config%vegetation_isolation_factor_urban  = cc%vegetation_isolation_factor_urban
    config%min_vegetation_fraction = cc%min_vegetation_fraction
    config%min_building_fraction   = cc%min_building_fraction
    config%iverbose = cc%iverbose
    ! The reference computes its own Legendre-Gauss tables
    ! (radsurf_config.F90:250-270)
    call config%consolidate()
  end subroutine fill_config

  subroutine copy_flux_out(f, cf)
    use radsurf_canopy_flux, only : canopy_flux_type
    type(canopy_flux_type), intent(in) :: f
    type(ss_flux), intent(in) :: cf
    integer :: ns, nc, nl
    ns = f%nspec; nc = f%ncol; nl = f%ntotlay
    call put_r2(f%ground_dn, cf%ground_dn, ns, nc)
    call put_r2(f%ground_dn_dir, cf%ground_dn_dir, ns, nc)
    call put_r2(f%ground_net, cf%ground_net, ns, nc)
    call put_r2(f%top_dn, cf%top_dn, ns, nc)
    call put_r2(f%top_dn_dir, cf%top_dn_dir, ns, nc)
    call put_r2(f%top_net, cf%top_net, ns, nc)
    call put_r2(f%roof_in, cf%roof_in, ns, nl)
    call put_r2(f%roof_in_dir, cf%roof_in_dir, ns, nl)
    call put_r2(f%roof_net, cf%roof_net, ns, nl)
    call put_r2(f%wall_in, cf%wall_in, ns, nl)
    call put_r2(f%wall_in_dir, cf%wall_in_dir, ns, nl)
    call put_r2(f%wall_net, cf%wall_net, ns, nl)
    call put_r1(f%roof_sunlit_frac, cf%roof_sunlit_frac, nl)
    call put_r1(f%wall_sunlit_frac, cf%wall_sunlit_frac, nl)
    call put_r1(f%ground_sunlit_frac, cf%ground_sunlit_frac, nc)
    call put_r2(f%ground_vertical_diff, cf%ground_vertical_diff, ns, nc)
    call put_r2(f%clear_air_abs, cf%clear_air_abs, ns, nl)
    call put_r2(f%veg_abs, cf%veg_abs, ns, nl)
    call put_r2(f%veg_air_abs, cf%veg_air_abs, ns, nl)
    call put_r2(f%veg_abs_dir, cf%veg_abs_dir, ns, nl)
    call put_r1(f%veg_sunlit_frac, cf%veg_sunlit_frac, nl)
    call put_r2(f%flux_dn_layer_top, cf%flux_dn_layer_top, ns, nl)
    call put_r2(f%flux_dn_dir_layer_top, cf%flux_dn_dir_layer_top, ns, nl)
    call put_r2(f%flux_up_layer_top, cf%flux_up_layer_top, ns, nl)
    call put_r2(f%flux_dn_layer_base, cf%flux_dn_layer_base, ns, nl)
    call put_r2(f%flux_dn_dir_layer_base, cf%flux_dn_dir_layer_base, ns, nl)
    call put_r2(f%flux_up_layer_base, cf%flux_up_layer_base, ns, nl)
  end subroutine copy_flux_out

  ! Entry with the argument list of spartacus_radsurf_hip plus the driver
  ! knobs.  do_simple_spectrum /= 0 runs calc_simple_spectrum_lw per block as
  ! the driver does; 0 uses the emission / Planck arrays exactly as passed.
  ! want_profiles selects the optional flux-profile arrays of canopy_flux.
  function spartacus_radsurf_ref(cc, ccan, csw, clw, cbc, istartcol_in, iendcol_in, &
       &  c_sw_dir, c_sw_diff, c_lw_int, c_lw_norm, &
       &  nblocksize, nrepeat, do_simple_spectrum, want_profiles, elapsed, &
       &  p_top_sw, p_top_sw_dir, p_top_lw, p_sw_flux, p_lw_flux) &
       &  bind(c, name=HARNESS_ENTRY) result(istatus)

    use omp_lib
    use radsurf_config,             only : config_type
    use radsurf_canopy_properties,  only : canopy_properties_type
    use radsurf_sw_spectral_properties, only : sw_spectral_properties_type
    use radsurf_lw_spectral_properties, only : lw_spectral_properties_type
    use radsurf_boundary_conds_out, only : boundary_conds_out_type
    use radsurf_canopy_flux,        only : canopy_flux_type
    use radsurf_interface,          only : radsurf
    use radsurf_simple_spectrum,    only : calc_simple_spectrum_lw

    type(ss_config),      intent(in) :: cc
    type(ss_canopy),      intent(in) :: ccan
    type(ss_sw_spectral), intent(in) :: csw
    type(ss_lw_spectral), intent(in) :: clw
    type(ss_bc_out),      intent(in) :: cbc
    integer(c_int32_t), value :: istartcol_in, iendcol_in
    type(ss_flux), intent(in) :: c_sw_dir, c_sw_diff, c_lw_int, c_lw_norm
    integer(c_int32_t), value :: nblocksize, nrepeat, do_simple_spectrum, want_profiles
    real(c_double), intent(out) :: elapsed
    ! Optional post-processing of the driver (spartacus_surface_driver.F90:250-261):
    ! non-null top-of-canopy fluxes (nsw,ncol)/(nlw,ncol) => canopy_flux%scale and
    ! %sum are applied; the scaled objects and the two sums are returned.
    type(c_ptr), value :: p_top_sw, p_top_sw_dir, p_top_lw, p_sw_flux, p_lw_flux
    integer(c_int) :: istatus
    real(c_double), allocatable :: top_sw(:,:), top_sw_dir(:,:), top_lw(:,:)
    type(canopy_flux_type) :: sw_flux, lw_flux
    type(ss_flux), pointer :: c_sw_flux, c_lw_flux

    type(config_type)                 :: config
    type(canopy_properties_type)      :: canopy_props
    type(sw_spectral_properties_type) :: sw_props
    type(lw_spectral_properties_type) :: lw_props
    type(boundary_conds_out_type)     :: bc_out
    type(canopy_flux_type) :: sw_norm_dir, sw_norm_diff, lw_internal, lw_norm

    integer :: ncol, ntotlay, nsw, nlw
    integer :: jrepeat, jblock, nblock, istartcol, iendcol
    real(c_double) :: tstart, tstop

    istatus = 0
    call fill_config(cc, config)

    ncol = ccan%ncol
    ntotlay = ccan%ntotlay
    nsw = csw%nspec
    nlw = clw%nspec

    canopy_props%ncol = ncol
    canopy_props%ntotlay = ntotlay
    call get_i1(ccan%nlay, ncol, canopy_props%nlay)
    call get_i1(ccan%istartlay, ncol, canopy_props%istartlay)
    call get_i1(ccan%i_representation, ncol, canopy_props%i_representation)
    call get_r1(ccan%cos_sza, ncol, canopy_props%cos_sza)
    call get_r1(ccan%dz, ntotlay, canopy_props%dz)
    call get_r1(ccan%building_fraction, ntotlay, canopy_props%building_fraction)
    call get_r1(ccan%veg_fraction, ntotlay, canopy_props%veg_fraction)
    call get_r1(ccan%building_scale, ntotlay, canopy_props%building_scale)
    call get_r1(ccan%veg_scale, ntotlay, canopy_props%veg_scale)
    call get_r1(ccan%veg_ext, ntotlay, canopy_props%veg_ext)
    call get_r1(ccan%veg_fsd, ntotlay, canopy_props%veg_fsd)
    call get_r1(ccan%veg_contact_fraction, ntotlay, canopy_props%veg_contact_fraction)
    call get_r1(ccan%ground_temperature, ncol, canopy_props%ground_temperature)
    call get_r1(ccan%roof_temperature, ntotlay, canopy_props%roof_temperature)
    call get_r1(ccan%wall_temperature, ntotlay, canopy_props%wall_temperature)
    call get_r1(ccan%clear_air_temperature, ntotlay, canopy_props%clear_air_temperature)
    call get_r1(ccan%veg_temperature, ntotlay, canopy_props%veg_temperature)
    call get_r1(ccan%veg_air_temperature, ntotlay, canopy_props%veg_air_temperature)

    sw_props%nspec = nsw; sw_props%ncol = ncol; sw_props%ntotlay = ntotlay
    call get_r2(csw%air_ext, nsw, ntotlay, sw_props%air_ext)
    call get_r2(csw%air_ssa, nsw, ntotlay, sw_props%air_ssa)
    call get_r2(csw%veg_ssa, nsw, ntotlay, sw_props%veg_ssa)
    call get_r2(csw%ground_albedo, nsw, ncol, sw_props%ground_albedo)
    call get_r2(csw%roof_albedo, nsw, ntotlay, sw_props%roof_albedo)
    call get_r2(csw%wall_albedo, nsw, ntotlay, sw_props%wall_albedo)
    call get_r2(csw%ground_albedo_dir, nsw, ncol, sw_props%ground_albedo_dir)
    call get_r2(csw%roof_albedo_dir, nsw, ntotlay, sw_props%roof_albedo_dir)
    call get_r2(csw%wall_specular_frac, nsw, ntotlay, sw_props%wall_specular_frac)

    lw_props%nspec = nlw; lw_props%ncol = ncol; lw_props%ntotlay = ntotlay
    call get_r2(clw%air_ext, nlw, ntotlay, lw_props%air_ext)
    call get_r2(clw%air_ssa, nlw, ntotlay, lw_props%air_ssa)
    call get_r2(clw%veg_ssa, nlw, ntotlay, lw_props%veg_ssa)
    call get_r2(clw%clear_air_planck, nlw, ntotlay, lw_props%clear_air_planck)
    call get_r2(clw%veg_planck, nlw, ntotlay, lw_props%veg_planck)
    call get_r2(clw%veg_air_planck, nlw, ntotlay, lw_props%veg_air_planck)
    call get_r2(clw%ground_emissivity, nlw, ncol, lw_props%ground_emissivity)
    call get_r2(clw%roof_emissivity, nlw, ntotlay, lw_props%roof_emissivity)
    call get_r2(clw%wall_emissivity, nlw, ntotlay, lw_props%wall_emissivity)
    call get_r2(clw%ground_emission, nlw, ncol, lw_props%ground_emission)
    call get_r2(clw%roof_emission, nlw, ntotlay, lw_props%roof_emission)
    call get_r2(clw%wall_emission, nlw, ntotlay, lw_props%wall_emission)

    ! Allocation pattern of the driver (spartacus_surface_driver.F90:166-184)
    call bc_out%allocate(ncol, config%nsw, config%nlw)
    bc_out%sw_albedo = 0.0_c_double
    bc_out%sw_albedo_dir = 0.0_c_double
    bc_out%lw_emissivity = 0.0_c_double
    bc_out%lw_emission = 0.0_c_double
    if (config%do_sw) then
      call sw_norm_dir%allocate(config, ncol, ntotlay, config%nsw, use_direct=.true., &
           &  do_save_flux_profile=(want_profiles /= 0))
      call sw_norm_diff%allocate(config, ncol, ntotlay, config%nsw, use_direct=.true., &
           &  do_save_flux_profile=(want_profiles /= 0))
      call sw_norm_dir%zero_all()
      call sw_norm_diff%zero_all()
    end if
    if (config%do_lw) then
      call lw_internal%allocate(config, ncol, ntotlay, config%nlw, use_direct=.false., &
           &  do_save_flux_profile=(want_profiles /= 0))
      call lw_norm%allocate(config, ncol, ntotlay, config%nlw, use_direct=.false., &
           &  do_save_flux_profile=(want_profiles /= 0))
      call lw_internal%zero_all()
      call lw_norm%zero_all()
    end if

    tstart = omp_get_wtime()
    do jrepeat = 1,max(1,nrepeat)
      nblock = (iendcol_in - istartcol_in + nblocksize) / nblocksize
      !$OMP PARALLEL DO PRIVATE(istartcol, iendcol) SCHEDULE(RUNTIME)
      do jblock = 1, nblock
        istartcol = (jblock-1) * nblocksize + istartcol_in
        iendcol = min(istartcol + nblocksize - 1, iendcol_in)
        if (config%do_lw .and. do_simple_spectrum /= 0) then
          call calc_simple_spectrum_lw(config, canopy_props, lw_props, &
               &                       istartcol, iendcol)
        end if
        call radsurf(config, canopy_props, sw_props, lw_props, bc_out, &
             &       istartcol, iendcol, sw_norm_dir, sw_norm_diff, &
             &       lw_internal, lw_norm)
      end do
      !$OMP END PARALLEL DO
    end do
    tstop = omp_get_wtime()
    elapsed = tstop - tstart

    if (config%do_sw .and. c_associated(p_top_sw) .and. c_associated(p_top_sw_dir) &
         &  .and. c_associated(p_sw_flux)) then
      call get_r2(p_top_sw, nsw, ncol, top_sw)
      call get_r2(p_top_sw_dir, nsw, ncol, top_sw_dir)
      call sw_flux%allocate(config, ncol, ntotlay, config%nsw, use_direct=.true., &
           &  do_save_flux_profile=(want_profiles /= 0))
      call sw_norm_dir%scale(canopy_props%nlay, top_sw_dir)
      call sw_norm_diff%scale(canopy_props%nlay, top_sw - top_sw_dir)
      call sw_flux%sum(sw_norm_dir, sw_norm_diff)
      call c_f_pointer(p_sw_flux, c_sw_flux)
      call copy_flux_out(sw_flux, c_sw_flux)
    end if
    if (config%do_lw .and. c_associated(p_top_lw) .and. c_associated(p_lw_flux)) then
      call get_r2(p_top_lw, nlw, ncol, top_lw)
      call lw_flux%allocate(config, ncol, ntotlay, config%nlw, use_direct=.false., &
           &  do_save_flux_profile=(want_profiles /= 0))
      call lw_norm%scale(canopy_props%nlay, top_lw)
      call lw_flux%sum(lw_internal, lw_norm)
      call c_f_pointer(p_lw_flux, c_lw_flux)
      call copy_flux_out(lw_flux, c_lw_flux)
    end if

    call put_r2(bc_out%sw_albedo, cbc%sw_albedo, config%nsw, ncol)
    call put_r2(bc_out%sw_albedo_dir, cbc%sw_albedo_dir, config%nsw, ncol)
    call put_r2(bc_out%lw_emissivity, cbc%lw_emissivity, config%nlw, ncol)
    call put_r2(bc_out%lw_emission, cbc%lw_emission, config%nlw, ncol)
    if (config%do_sw) then
      call copy_flux_out(sw_norm_dir, c_sw_dir)
      call copy_flux_out(sw_norm_diff, c_sw_diff)
    end if
    if (config%do_lw) then
      call copy_flux_out(lw_internal, c_lw_int)
      call copy_flux_out(lw_norm, c_lw_norm)
      ! Emission arrays computed by calc_simple_spectrum_lw go back too
      if (do_simple_spectrum /= 0) then
        call put_r2(lw_props%clear_air_planck, clw%clear_air_planck, nlw, ntotlay)
        call put_r2(lw_props%veg_planck, clw%veg_planck, nlw, ntotlay)
        call put_r2(lw_props%veg_air_planck, clw%veg_air_planck, nlw, ntotlay)
        call put_r2(lw_props%ground_emission, clw%ground_emission, nlw, ncol)
        call put_r2(lw_props%roof_emission, clw%roof_emission, nlw, ntotlay)
        call put_r2(lw_props%wall_emission, clw%wall_emission, nlw, ntotlay)
      end if
    end if

  end function spartacus_radsurf_ref


  ! Return the reference's own Legendre-Gauss table for nstream streams
  ! (radtool_legendre_gauss.F90:52-100)
  subroutine ref_legendre_gauss(nstream, clg) bind(c, name='ref_legendre_gauss')
    use radtool_legendre_gauss, only : legendre_gauss_type
    integer(c_int32_t), value :: nstream
    type(ss_legendre_gauss), intent(out) :: clg
    type(legendre_gauss_type) :: lg
    call lg%initialize(nstream)
    clg%nstream = nstream
    clg%pad_ = 0
    clg%mu = 0; clg%sin_ang = 0; clg%tan_ang = 0
    clg%weight = 0; clg%hweight = 0; clg%vweight = 0
    clg%mu(1:nstream) = lg%mu
    clg%sin_ang(1:nstream) = lg%sin_ang
    clg%tan_ang(1:nstream) = lg%tan_ang
    clg%weight(1:nstream) = lg%weight
    clg%hweight(1:nstream) = lg%hweight
    clg%vweight(1:nstream) = lg%vweight
    clg%vadjustment = lg%vadjustment
    clg%vadjustment2 = lg%vadjustment2
  end subroutine ref_legendre_gauss


  ! Kernel-level entries: the reference's eigen decomposition and layer-matrix
  ! routines on "nmat" stacked matrices, Fortran layout (nmat, i, j).
  subroutine ref_eigen_decomposition_real(norder, nmat, amat, eigenvalue, eigenvector) &
       &  bind(c, name='ref_eigen_decomposition_real')
    use radtool_eigen_decomposition, only : eigen_decomposition_real
    integer(c_int32_t), value :: norder, nmat
    real(c_double), intent(in)  :: amat(nmat,norder,norder)
    real(c_double), intent(out) :: eigenvalue(nmat,norder)
    real(c_double), intent(out) :: eigenvector(nmat,norder,norder)
    call eigen_decomposition_real(norder, nmat, amat, eigenvalue, eigenvector)
  end subroutine ref_eigen_decomposition_real

  subroutine ref_calc_matrices_sw_eig(nmat, ndiff, ndir, dz, mu0, &
       &  gamma0, gamma1, gamma2, gamma3, &
       &  reflectance, transmittance, s_up, s_dn, trans_dir, &
       &  int_dir, int_diff, int_dir_diff) bind(c, name='ref_calc_matrices_sw_eig')
    use radtool_calc_matrices_sw_eig, only : calc_matrices_sw_eig
    integer(c_int32_t), value :: nmat, ndiff, ndir
    real(c_double), value :: dz, mu0
    real(c_double), intent(in) :: gamma0(nmat,ndir,ndir), gamma1(nmat,ndiff,ndiff)
    real(c_double), intent(in) :: gamma2(nmat,ndiff,ndiff), gamma3(nmat,ndiff,ndir)
    real(c_double), intent(out) :: reflectance(nmat,ndiff,ndiff), transmittance(nmat,ndiff,ndiff)
    real(c_double), intent(out) :: s_up(nmat,ndiff,ndir), s_dn(nmat,ndiff,ndir)
    real(c_double), intent(out) :: trans_dir(nmat,ndir,ndir)
    real(c_double), intent(out) :: int_dir(nmat,ndir,ndir), int_diff(nmat,ndiff,ndiff)
    real(c_double), intent(out) :: int_dir_diff(nmat,ndiff,ndir)
    call calc_matrices_sw_eig(nmat, ndiff, ndir, dz, mu0, gamma0, gamma1, gamma2, gamma3, &
         &  reflectance, transmittance, s_up, s_dn, trans_dir, int_dir, int_diff, int_dir_diff)
  end subroutine ref_calc_matrices_sw_eig

  subroutine ref_calc_matrices_lw_eig(nmat, norder, dz, gamma1, gamma2, emiss_rate, &
       &  reflectance, transmittance, source, int_flux, int_flux_source) &
       &  bind(c, name='ref_calc_matrices_lw_eig')
    use radtool_calc_matrices_lw_eig, only : calc_matrices_lw_eig
    integer(c_int32_t), value :: nmat, norder
    real(c_double), value :: dz
    real(c_double), intent(in) :: gamma1(nmat,norder,norder), gamma2(nmat,norder,norder)
    real(c_double), intent(in) :: emiss_rate(nmat,norder)
    real(c_double), intent(out) :: reflectance(nmat,norder,norder), transmittance(nmat,norder,norder)
    real(c_double), intent(out) :: source(nmat,norder)
    real(c_double), intent(out) :: int_flux(nmat,norder,norder), int_flux_source(nmat,norder)
    call calc_matrices_lw_eig(nmat, norder, dz, gamma1, gamma2, emiss_rate, &
         &  reflectance, transmittance, source, int_flux, int_flux_source)
  end subroutine ref_calc_matrices_lw_eig

end module ref_harness
