! radsurf_interface_hip.F90 - drop-in replacement for the reference's
! "module radsurf_interface" (radsurf/radsurf_interface.F90:16-315) that runs the
! per-column SW/LW flux solver on an AMD MI355X through the C-ABI of
! include/spartacus_surface_hip.h.
!
! Link this object IN PLACE OF radsurf_interface.o.  The subroutine keeps the
! reference's name, argument names, order, types, intents and optional
! attributes (radsurf_interface.F90:20-59), so the offline driver
! (driver/spartacus_surface_driver.F90:228-231), read_input, save_canopy_fluxes
! and canopy_flux%scale/sum are untouched.  The only change to the interface is
! the TARGET attribute on the dummies, needed to take c_loc() of their
! allocatable components (the reference already has it on the two spectral
! property arguments).
!
! All derived types stay the reference's own (config_type,
! canopy_properties_type, sw/lw_spectral_properties_type,
! boundary_conds_out_type, canopy_flux_type): this file only builds flat
! bind(C) views {dimensions, c_ptr to each allocatable component or
! c_null_ptr when it is not allocated} and hands them to
! spartacus_radsurf_hip.  No copies are made on the Fortran side.
!
! Errors: a non-zero status from the library is turned into radiation_abort
! (utilities/radiation_io.F90:46-54), which is what the reference does for its
! own fatal conditions (radsurf_interface.F90:281-284, radsurf_urban_sw.F90:275-277).
!
! Threading: the reference driver calls radsurf concurrently from OpenMP
! threads on disjoint column ranges (driver :206-233).  The library serialises
! such calls; for throughput set nblocksize >= ncol in &radsurf_driver so that
! one call covers all columns.

module radsurf_interface

  use, intrinsic :: iso_c_binding

  implicit none

  private
  public :: radsurf

  integer, parameter :: SS_MAX_STREAMS = 16

  ! ---- bind(C) mirrors of the structs in include/spartacus_surface_hip.h ----
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

  interface
    function spartacus_radsurf_hip(config, canopy, sw, lw, bc_out, istartcol, iendcol, &
         &  sw_norm_dir, sw_norm_diff, lw_internal, lw_norm) bind(c, name='spartacus_radsurf_hip')
      import :: c_int, c_int32_t, c_ptr, ss_config, ss_canopy, ss_sw_spectral, ss_lw_spectral, ss_bc_out
      type(ss_config), intent(in)      :: config
      type(ss_canopy), intent(in)      :: canopy
      type(ss_sw_spectral), intent(in) :: sw
      type(ss_lw_spectral), intent(in) :: lw
      type(ss_bc_out), intent(in)      :: bc_out
      integer(c_int32_t), value        :: istartcol, iendcol
      type(c_ptr), value               :: sw_norm_dir, sw_norm_diff, lw_internal, lw_norm
      integer(c_int)                   :: spartacus_radsurf_hip
    end function spartacus_radsurf_hip
    function ss_last_error() bind(c, name='ss_last_error')
      import :: c_ptr
      type(c_ptr) :: ss_last_error
    end function ss_last_error
  end interface

  ! address of an allocatable component, c_null_ptr if not allocated
  interface cptr
    module procedure cptr_r1, cptr_r2, cptr_i1
  end interface cptr

contains

  function cptr_r1(a) result(p)
    use parkind1, only : jprb
    real(kind=jprb), allocatable, target, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function cptr_r1

  function cptr_r2(a) result(p)
    use parkind1, only : jprb
    real(kind=jprb), allocatable, target, intent(in) :: a(:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function cptr_r2

  function cptr_i1(a) result(p)
    use parkind1, only : jpim
    integer(kind=jpim), allocatable, target, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function cptr_i1

  subroutine fill_lg(lg, clg)
    use radtool_legendre_gauss, only : legendre_gauss_type
    type(legendre_gauss_type), intent(in) :: lg
    type(ss_legendre_gauss), intent(out)  :: clg
    integer :: n
    n = lg%nstream
    clg%nstream = n
    clg%pad_ = 0
    clg%mu = 0.0_c_double;      clg%sin_ang = 0.0_c_double; clg%tan_ang = 0.0_c_double
    clg%weight = 0.0_c_double;  clg%hweight = 0.0_c_double; clg%vweight = 0.0_c_double
    clg%vadjustment = 1.0_c_double
    clg%vadjustment2 = 1.0_c_double
    if (n > 0 .and. n <= SS_MAX_STREAMS) then
      clg%mu(1:n)      = lg%mu
      clg%sin_ang(1:n) = lg%sin_ang
      clg%tan_ang(1:n) = lg%tan_ang
      clg%weight(1:n)  = lg%weight
      clg%hweight(1:n) = lg%hweight
      clg%vweight(1:n) = lg%vweight
      clg%vadjustment  = lg%vadjustment
      clg%vadjustment2 = lg%vadjustment2
    end if
  end subroutine fill_lg

  subroutine fill_flux(f, cf)
    use radsurf_canopy_flux, only : canopy_flux_type
    type(canopy_flux_type), target, intent(in) :: f
    type(ss_flux), intent(out) :: cf
    cf%nspec = f%nspec; cf%ncol = f%ncol; cf%ntotlay = f%ntotlay; cf%pad_ = 0
    cf%ground_dn = cptr(f%ground_dn);           cf%ground_dn_dir = cptr(f%ground_dn_dir)
    cf%ground_net = cptr(f%ground_net);         cf%top_dn = cptr(f%top_dn)
    cf%top_dn_dir = cptr(f%top_dn_dir);         cf%top_net = cptr(f%top_net)
    cf%roof_in = cptr(f%roof_in);               cf%roof_in_dir = cptr(f%roof_in_dir)
    cf%roof_net = cptr(f%roof_net);             cf%wall_in = cptr(f%wall_in)
    cf%wall_in_dir = cptr(f%wall_in_dir);       cf%wall_net = cptr(f%wall_net)
    cf%roof_sunlit_frac = cptr(f%roof_sunlit_frac)
    cf%wall_sunlit_frac = cptr(f%wall_sunlit_frac)
    cf%ground_sunlit_frac = cptr(f%ground_sunlit_frac)
    cf%ground_vertical_diff = cptr(f%ground_vertical_diff)
    cf%clear_air_abs = cptr(f%clear_air_abs);   cf%veg_abs = cptr(f%veg_abs)
    cf%veg_air_abs = cptr(f%veg_air_abs);       cf%veg_abs_dir = cptr(f%veg_abs_dir)
    cf%veg_sunlit_frac = cptr(f%veg_sunlit_frac)
    cf%flux_dn_layer_top = cptr(f%flux_dn_layer_top)
    cf%flux_dn_dir_layer_top = cptr(f%flux_dn_dir_layer_top)
    cf%flux_up_layer_top = cptr(f%flux_up_layer_top)
    cf%flux_dn_layer_base = cptr(f%flux_dn_layer_base)
    cf%flux_dn_dir_layer_base = cptr(f%flux_dn_dir_layer_base)
    cf%flux_up_layer_base = cptr(f%flux_up_layer_base)
  end subroutine fill_flux

  pure integer(c_int32_t) function l2i(l)
    logical, intent(in) :: l
    if (l) then
      l2i = 1
    else
      l2i = 0
    end if
  end function l2i

  ! Same name, arguments and meaning as the reference's radsurf
  ! (radsurf/radsurf_interface.F90:20-25)
  subroutine radsurf(config, canopy_props, &
       &             sw_spectral_props, lw_spectral_props, &
       &             bc_out, &
       &             istartcol, iendcol, &
       &             sw_norm_dir, sw_norm_diff, &
       &             lw_internal, lw_norm)

    use parkind1,                   only : jpim, jprb
    use radiation_io,               only : nulerr, radiation_abort
    use radsurf_config,             only : config_type
    use radsurf_canopy_properties,  only : canopy_properties_type
    use radsurf_sw_spectral_properties, only : sw_spectral_properties_type
    use radsurf_lw_spectral_properties, only : lw_spectral_properties_type
    use radsurf_boundary_conds_out, only : boundary_conds_out_type
    use radsurf_canopy_flux,        only : canopy_flux_type

    type(config_type),             intent(in)  :: config
    type(canopy_properties_type),  intent(in), target :: canopy_props
    type(lw_spectral_properties_type), intent(in), target :: lw_spectral_props
    type(sw_spectral_properties_type), intent(in), target :: sw_spectral_props
    type(boundary_conds_out_type), intent(inout), target :: bc_out
    type(canopy_flux_type),        intent(inout), optional, target &
         &  :: sw_norm_dir, &  ! SW fluxes normalized by top-of-canopy direct
         &     sw_norm_diff, & ! SW fluxes normalized by top-of-canopy diffuse
         &     lw_internal, &  ! LW fluxes from internal emission
         &     lw_norm         ! LW fluxes normalized by top-of-canopy down
    integer(kind=jpim), optional, intent(in) :: istartcol, iendcol

    type(ss_config)      :: cc
    type(ss_canopy)      :: ccan
    type(ss_sw_spectral) :: csw
    type(ss_lw_spectral) :: clw
    type(ss_bc_out)      :: cbc
    type(ss_flux), target :: cf(4)
    type(c_ptr) :: pf(4)
    integer(kind=jpim) :: icol1, icol2
    integer(c_int) :: istatus
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: pmsg
    integer :: jc

    if (kind(1.0_jprb) /= c_double) then
      call radiation_abort('radsurf (HIP): the MI355X solver is fp64; build without SINGLE_PRECISION')
    end if

    ! Range of columns to process (radsurf_interface.F90:86-96)
    if (present(istartcol)) then
      icol1 = istartcol
    else
      icol1 = 1
    end if
    if (present(iendcol)) then
      icol2 = iendcol
    else
      icol2 = canopy_props%ncol
    end if

    cc%do_sw = l2i(config%do_sw)
    cc%do_lw = l2i(config%do_lw)
    cc%use_sw_direct_albedo = l2i(config%use_sw_direct_albedo)
    cc%do_vegetation = l2i(config%do_vegetation)
    cc%do_urban = l2i(config%do_urban)
    cc%n_vegetation_region_forest = config%n_vegetation_region_forest
    cc%n_vegetation_region_urban  = config%n_vegetation_region_urban
    cc%nsw = config%nswinternal
    cc%nlw = config%nlwinternal
    cc%use_symmetric_vegetation_scale_forest = l2i(config%use_symmetric_vegetation_scale_forest)
    cc%use_symmetric_vegetation_scale_urban  = l2i(config%use_symmetric_vegetation_scale_urban)
    cc%iverbose = config%iverbose
    cc%vegetation_isolation_factor_forest = config%vegetation_isolation_factor_forest
    cc%vegetation_isolation_factor_urban  = config%vegetation_isolation_factor_urban
    cc%min_vegetation_fraction = config%min_vegetation_fraction
    cc%min_building_fraction   = config%min_building_fraction
    call fill_lg(config%lg_sw_forest, cc%lg_sw_forest)
    call fill_lg(config%lg_sw_urban,  cc%lg_sw_urban)
    call fill_lg(config%lg_lw_forest, cc%lg_lw_forest)
    call fill_lg(config%lg_lw_urban,  cc%lg_lw_urban)

    ccan%ncol = canopy_props%ncol
    ccan%ntotlay = canopy_props%ntotlay
    ccan%nlay = cptr(canopy_props%nlay)
    ccan%istartlay = cptr(canopy_props%istartlay)
    ccan%i_representation = cptr(canopy_props%i_representation)
    ccan%cos_sza = cptr(canopy_props%cos_sza)
    ccan%dz = cptr(canopy_props%dz)
    ccan%building_fraction = cptr(canopy_props%building_fraction)
    ccan%veg_fraction = cptr(canopy_props%veg_fraction)
    ccan%building_scale = cptr(canopy_props%building_scale)
    ccan%veg_scale = cptr(canopy_props%veg_scale)
    ccan%veg_ext = cptr(canopy_props%veg_ext)
    ccan%veg_fsd = cptr(canopy_props%veg_fsd)
    ccan%veg_contact_fraction = cptr(canopy_props%veg_contact_fraction)
    ccan%ground_temperature = cptr(canopy_props%ground_temperature)
    ccan%roof_temperature = cptr(canopy_props%roof_temperature)
    ccan%wall_temperature = cptr(canopy_props%wall_temperature)
    ccan%clear_air_temperature = cptr(canopy_props%clear_air_temperature)
    ccan%veg_temperature = cptr(canopy_props%veg_temperature)
    ccan%veg_air_temperature = cptr(canopy_props%veg_air_temperature)

    csw%nspec = config%nswinternal
    csw%pad_ = 0
    csw%air_ext = cptr(sw_spectral_props%air_ext)
    csw%air_ssa = cptr(sw_spectral_props%air_ssa)
    csw%veg_ssa = cptr(sw_spectral_props%veg_ssa)
    csw%ground_albedo = cptr(sw_spectral_props%ground_albedo)
    csw%roof_albedo = cptr(sw_spectral_props%roof_albedo)
    csw%wall_albedo = cptr(sw_spectral_props%wall_albedo)
    csw%ground_albedo_dir = cptr(sw_spectral_props%ground_albedo_dir)
    csw%roof_albedo_dir = cptr(sw_spectral_props%roof_albedo_dir)
    csw%wall_specular_frac = cptr(sw_spectral_props%wall_specular_frac)

    clw%nspec = config%nlwinternal
    clw%pad_ = 0
    clw%air_ext = cptr(lw_spectral_props%air_ext)
    clw%air_ssa = cptr(lw_spectral_props%air_ssa)
    clw%veg_ssa = cptr(lw_spectral_props%veg_ssa)
    clw%clear_air_planck = cptr(lw_spectral_props%clear_air_planck)
    clw%veg_planck = cptr(lw_spectral_props%veg_planck)
    clw%veg_air_planck = cptr(lw_spectral_props%veg_air_planck)
    clw%ground_emissivity = cptr(lw_spectral_props%ground_emissivity)
    clw%roof_emissivity = cptr(lw_spectral_props%roof_emissivity)
    clw%wall_emissivity = cptr(lw_spectral_props%wall_emissivity)
    clw%ground_emission = cptr(lw_spectral_props%ground_emission)
    clw%roof_emission = cptr(lw_spectral_props%roof_emission)
    clw%wall_emission = cptr(lw_spectral_props%wall_emission)

    cbc%sw_albedo = cptr(bc_out%sw_albedo)
    cbc%sw_albedo_dir = cptr(bc_out%sw_albedo_dir)
    cbc%lw_emissivity = cptr(bc_out%lw_emissivity)
    cbc%lw_emission = cptr(bc_out%lw_emission)

    pf = c_null_ptr
    if (present(sw_norm_dir)) then
      call fill_flux(sw_norm_dir, cf(1));  pf(1) = c_loc(cf(1))
    end if
    if (present(sw_norm_diff)) then
      call fill_flux(sw_norm_diff, cf(2)); pf(2) = c_loc(cf(2))
    end if
    if (present(lw_internal)) then
      call fill_flux(lw_internal, cf(3));  pf(3) = c_loc(cf(3))
    end if
    if (present(lw_norm)) then
      call fill_flux(lw_norm, cf(4));      pf(4) = c_loc(cf(4))
    end if

    istatus = spartacus_radsurf_hip(cc, ccan, csw, clw, cbc, &
         &  int(icol1, c_int32_t), int(icol2, c_int32_t), pf(1), pf(2), pf(3), pf(4))

    if (istatus /= 0) then
      pmsg = ss_last_error()
      if (c_associated(pmsg)) then
        call c_f_pointer(pmsg, msg, [256])
        write(nulerr,'(a)',advance='no') '*** Error in spartacus_radsurf_hip: '
        do jc = 1,256
          if (msg(jc) == c_null_char) exit
          write(nulerr,'(a)',advance='no') msg(jc)
        end do
        write(nulerr,'(a)') ''
      end if
      write(nulerr,'(a,i0)') '*** spartacus_radsurf_hip returned status ', istatus
      call radiation_abort('MI355X column solver failed')
    end if

  end subroutine radsurf

end module radsurf_interface
