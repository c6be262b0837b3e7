﻿!mod$ v1 sum:3e24925acc0be69d
!need$ 7e14fb4b785524cf n radtool_legendre_gauss
!need$ a63c8333ed1677c5 n parkind1
module radsurf_config
use parkind1,only:jpim
use parkind1,only:jprb
use radtool_legendre_gauss,only:legendre_gauss_type
integer(4),parameter::nprintstringlen=41_4
type::config_type
logical(4)::do_sw=.true._4
logical(4)::do_lw=.true._4
logical(4)::use_sw_direct_albedo=.false._4
logical(4)::do_vegetation=.true._4
logical(4)::do_urban=.true._4
integer(4)::n_vegetation_region_forest=1_4
integer(4)::n_vegetation_region_urban=1_4
integer(4)::nsw=1_4
integer(4)::nlw=1_4
integer(4)::n_stream_sw_forest=4_4
integer(4)::n_stream_sw_urban=4_4
integer(4)::n_stream_lw_forest=4_4
integer(4)::n_stream_lw_urban=4_4
logical(4)::use_symmetric_vegetation_scale_forest=.true._4
logical(4)::use_symmetric_vegetation_scale_urban=.true._4
real(8)::vegetation_isolation_factor_forest=0._8
real(8)::vegetation_isolation_factor_urban=0._8
real(8)::min_vegetation_fraction=9.999999974752427078783512115478515625e-7_8
real(8)::min_building_fraction=9.999999974752427078783512115478515625e-7_8
logical(4)::do_save_broadband_flux=.true._4
logical(4)::do_save_spectral_flux=.false._4
logical(4)::do_save_flux_profile=.false._4
integer(4)::iverbose=3_4
integer(4)::nswinternal
integer(4)::nlwinternal
type(legendre_gauss_type)::lg_sw_urban
type(legendre_gauss_type)::lg_sw_forest
type(legendre_gauss_type)::lg_lw_urban
type(legendre_gauss_type)::lg_lw_forest
contains
procedure::read=>read_config_from_namelist
procedure::consolidate=>consolidate_config
procedure::print=>print_config
end type
contains
subroutine read_config_from_namelist(this,file_name,unit,is_success)
class(config_type),intent(inout),target::this
character(*,1),intent(in),optional::file_name
integer(4),intent(in),optional::unit
logical(4),intent(out),optional::is_success
end
subroutine consolidate_config(this)
class(config_type),intent(inout)::this
end
subroutine print_config(this,iverbose)
class(config_type),intent(in)::this
integer(4),intent(in),optional::iverbose
end
subroutine print_logical(message_str,name,val)
character(*,1),intent(in)::message_str
character(*,1),intent(in)::name
logical(4),intent(in)::val
end
subroutine print_integer(message_str,name,val)
character(*,1),intent(in)::message_str
character(*,1),intent(in)::name
integer(4),intent(in)::val
end
subroutine print_real(message_str,name,val)
character(*,1),intent(in)::message_str
character(*,1),intent(in)::name
real(8),intent(in)::val
end
end
