﻿!mod$ v1 sum:dde65d117102be91
!need$ a7fbb53af6e3c6a5 n ref_harness_types
!need$ bb381bf46e508468 i __fortran_builtins
!need$ 0bde2ac47243ead2 i iso_c_binding
module ref_harness
use ref_harness_types,only:ss_max_streams
use ref_harness_types,only:ss_legendre_gauss
use ref_harness_types,only:ss_config
use ref_harness_types,only:ss_canopy
use ref_harness_types,only:ss_sw_spectral
use ref_harness_types,only:ss_lw_spectral
use ref_harness_types,only:ss_bc_out
use ref_harness_types,only:ss_flux
use ref_harness_types,only:get_r1
use ref_harness_types,only:get_r2
use ref_harness_types,only:get_i1
use ref_harness_types,only:put_r1
use ref_harness_types,only:put_r2
use,intrinsic::iso_c_binding,only:c_associated
use,intrinsic::iso_c_binding,only:c_funloc
use,intrinsic::iso_c_binding,only:c_funptr
use,intrinsic::iso_c_binding,only:c_f_pointer
use,intrinsic::iso_c_binding,only:c_loc
use,intrinsic::iso_c_binding,only:c_null_funptr
use,intrinsic::iso_c_binding,only:c_null_ptr
use,intrinsic::iso_c_binding,only:c_ptr
use,intrinsic::iso_c_binding,only:c_sizeof
use,intrinsic::iso_c_binding,only:operator(==)
use,intrinsic::iso_c_binding,only:operator(/=)
use,intrinsic::iso_c_binding,only:c_int8_t
use,intrinsic::iso_c_binding,only:c_int16_t
use,intrinsic::iso_c_binding,only:c_int32_t
use,intrinsic::iso_c_binding,only:c_int64_t
use,intrinsic::iso_c_binding,only:c_int128_t
use,intrinsic::iso_c_binding,only:c_int
use,intrinsic::iso_c_binding,only:c_short
use,intrinsic::iso_c_binding,only:c_long
use,intrinsic::iso_c_binding,only:c_long_long
use,intrinsic::iso_c_binding,only:c_signed_char
use,intrinsic::iso_c_binding,only:c_size_t
use,intrinsic::iso_c_binding,only:c_intmax_t
use,intrinsic::iso_c_binding,only:c_intptr_t
use,intrinsic::iso_c_binding,only:c_ptrdiff_t
use,intrinsic::iso_c_binding,only:c_int_least8_t
use,intrinsic::iso_c_binding,only:c_int_fast8_t
use,intrinsic::iso_c_binding,only:c_int_least16_t
use,intrinsic::iso_c_binding,only:c_int_fast16_t
use,intrinsic::iso_c_binding,only:c_int_least32_t
use,intrinsic::iso_c_binding,only:c_int_fast32_t
use,intrinsic::iso_c_binding,only:c_int_least64_t
use,intrinsic::iso_c_binding,only:c_int_fast64_t
use,intrinsic::iso_c_binding,only:c_int_least128_t
use,intrinsic::iso_c_binding,only:c_int_fast128_t
use,intrinsic::iso_c_binding,only:c_float
use,intrinsic::iso_c_binding,only:c_double
use,intrinsic::iso_c_binding,only:c_long_double
use,intrinsic::iso_c_binding,only:c_float_complex
use,intrinsic::iso_c_binding,only:c_double_complex
use,intrinsic::iso_c_binding,only:c_long_double_complex
use,intrinsic::iso_c_binding,only:c_bool
use,intrinsic::iso_c_binding,only:c_char
use,intrinsic::iso_c_binding,only:c_null_char
use,intrinsic::iso_c_binding,only:c_alert
use,intrinsic::iso_c_binding,only:c_backspace
use,intrinsic::iso_c_binding,only:c_form_feed
use,intrinsic::iso_c_binding,only:c_new_line
use,intrinsic::iso_c_binding,only:c_carriage_return
use,intrinsic::iso_c_binding,only:c_horizontal_tab
use,intrinsic::iso_c_binding,only:c_vertical_tab
use,intrinsic::iso_c_binding,only:c_float128
use,intrinsic::iso_c_binding,only:c_float128_complex
use,intrinsic::iso_c_binding,only:c_uint8_t
use,intrinsic::iso_c_binding,only:c_uint16_t
use,intrinsic::iso_c_binding,only:c_uint32_t
use,intrinsic::iso_c_binding,only:c_uint64_t
use,intrinsic::iso_c_binding,only:c_uint128_t
use,intrinsic::iso_c_binding,only:c_unsigned_char
use,intrinsic::iso_c_binding,only:c_unsigned_short
use,intrinsic::iso_c_binding,only:c_unsigned
use,intrinsic::iso_c_binding,only:c_unsigned_long
use,intrinsic::iso_c_binding,only:c_unsigned_long_long
use,intrinsic::iso_c_binding,only:c_uintmax_t
use,intrinsic::iso_c_binding,only:c_uint_fast8_t
use,intrinsic::iso_c_binding,only:c_uint_fast16_t
use,intrinsic::iso_c_binding,only:c_uint_fast32_t
use,intrinsic::iso_c_binding,only:c_uint_fast64_t
use,intrinsic::iso_c_binding,only:c_uint_fast128_t
use,intrinsic::iso_c_binding,only:c_uint_least8_t
use,intrinsic::iso_c_binding,only:c_uint_least16_t
use,intrinsic::iso_c_binding,only:c_uint_least32_t
use,intrinsic::iso_c_binding,only:c_uint_least64_t
use,intrinsic::iso_c_binding,only:c_uint_least128_t
use,intrinsic::iso_c_binding,only:c_f_procpointer
use,intrinsic::__fortran_builtins,only:iso_c_binding$__fortran_builtins$c_associated_c_ptr=>c_associated_c_ptr
private::iso_c_binding$__fortran_builtins$c_associated_c_ptr
contains
subroutine fill_config(cc,config)
use radsurf_config,only:config_type
type(ss_config),intent(in)::cc
type(config_type),intent(inout)::config
end
subroutine copy_flux_out(f,cf)
use radsurf_canopy_flux,only:canopy_flux_type
type(canopy_flux_type),intent(in)::f
type(ss_flux),intent(in)::cf
end
function spartacus_radsurf_ref(cc,ccan,csw,clw,cbc,istartcol_in,iendcol_in,c_sw_dir,c_sw_diff,c_lw_int,c_lw_norm,nblocksize,nrepeat,do_simple_spectrum,want_profiles,elapsed,p_top_sw,p_top_sw_dir,p_top_lw,p_sw_flux,p_lw_flux) bind(c,name="spartacus_radsurf_ref") result(istatus)
type(ss_config),intent(in)::cc
type(ss_canopy),intent(in)::ccan
type(ss_sw_spectral),intent(in)::csw
type(ss_lw_spectral),intent(in)::clw
type(ss_bc_out),intent(in)::cbc
integer(4),value::istartcol_in
integer(4),value::iendcol_in
type(ss_flux),intent(in)::c_sw_dir
type(ss_flux),intent(in)::c_sw_diff
type(ss_flux),intent(in)::c_lw_int
type(ss_flux),intent(in)::c_lw_norm
integer(4),value::nblocksize
integer(4),value::nrepeat
integer(4),value::do_simple_spectrum
integer(4),value::want_profiles
real(8),intent(out)::elapsed
type(c_ptr),value::p_top_sw
type(c_ptr),value::p_top_sw_dir
type(c_ptr),value::p_top_lw
type(c_ptr),value::p_sw_flux
type(c_ptr),value::p_lw_flux
integer(4)::istatus
end
subroutine ref_legendre_gauss(nstream,clg) bind(c,name="ref_legendre_gauss")
integer(4),value::nstream
type(ss_legendre_gauss),intent(out)::clg
end
subroutine ref_eigen_decomposition_real(norder,nmat,amat,eigenvalue,eigenvector) bind(c,name="ref_eigen_decomposition_real")
integer(4),value::norder
integer(4),value::nmat
real(8),intent(in)::amat(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::eigenvalue(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::eigenvector(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
end
subroutine ref_calc_matrices_sw_eig(nmat,ndiff,ndir,dz,mu0,gamma0,gamma1,gamma2,gamma3,reflectance,transmittance,s_up,s_dn,trans_dir,int_dir,int_diff,int_dir_diff) bind(c,name="ref_calc_matrices_sw_eig")
integer(4),value::nmat
integer(4),value::ndiff
integer(4),value::ndir
real(8),value::dz
real(8),value::mu0
real(8),intent(in)::gamma0(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8),1_8:int(ndir,kind=8))
real(8),intent(in)::gamma1(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(in)::gamma2(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(in)::gamma3(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::reflectance(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(out)::transmittance(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(out)::s_up(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::s_dn(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::trans_dir(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::int_dir(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::int_diff(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(out)::int_dir_diff(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
end
subroutine ref_calc_matrices_lw_eig(nmat,norder,dz,gamma1,gamma2,emiss_rate,reflectance,transmittance,source,int_flux,int_flux_source) bind(c,name="ref_calc_matrices_lw_eig")
integer(4),value::nmat
integer(4),value::norder
real(8),value::dz
real(8),intent(in)::gamma1(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(in)::gamma2(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(in)::emiss_rate(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::reflectance(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::transmittance(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::source(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::int_flux(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::int_flux_source(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
end
end
