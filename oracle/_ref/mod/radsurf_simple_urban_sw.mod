﻿!mod$ v1 sum:a6101585238b27f8
module radsurf_simple_urban_sw
contains
subroutine simple_urban_sw(config,is_infinite_street,nsw,icol,ilay,cos_sza,canopy_props,sw_spectral_props,ground_albedo_diff,ground_albedo_dir,top_albedo_diff,top_albedo_dir,sw_norm_dir,sw_norm_diff)
use radsurf_canopy_flux,only:canopy_flux_type
use radsurf_canopy_properties,only:canopy_properties_type
use radsurf_config,only:config_type
use radsurf_sw_spectral_properties,only:sw_spectral_properties_type
type(config_type),intent(in)::config
logical(4),intent(in)::is_infinite_street
integer(4),intent(in)::nsw
integer(4),intent(in)::icol
integer(4),intent(in)::ilay
real(8),intent(in)::cos_sza
type(canopy_properties_type),intent(in)::canopy_props
type(sw_spectral_properties_type),intent(in)::sw_spectral_props
real(8),intent(in)::ground_albedo_diff(1_8:int(nsw,kind=8))
real(8),intent(in)::ground_albedo_dir(1_8:int(nsw,kind=8))
real(8),intent(out)::top_albedo_diff(1_8:int(nsw,kind=8))
real(8),intent(out)::top_albedo_dir(1_8:int(nsw,kind=8))
type(canopy_flux_type),intent(inout),optional::sw_norm_dir
type(canopy_flux_type),intent(inout),optional::sw_norm_diff
end
end
